"""Alias package: put `<repo>/busca_amd/compat` on PYTHONPATH and the trackers' unchanged imports
(`from busca.network import BUSCA`, `from busca.tracking import center_distance`,
`from busca.option import load_args_from_config, merge_args`, `from busca.visualization import plot_box`)
resolve to the MI355X implementation in busca_amd."""
