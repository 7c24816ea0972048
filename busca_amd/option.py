"""YAML config -> namespaces, same contract as the reference's busca/option.py:10-39
(`load_args_from_config` returns (tracker_args, trainer_args), both carrying `.transformer`;
`merge_args` overlays non-None values of `new_args`)."""
import copy
import types

import yaml


def load_args_from_config(config_file):
    with open(config_file, "r") as fh:
        cfg = yaml.safe_load(fh)
    ns = {sec: types.SimpleNamespace(**cfg[sec]) for sec in ("tracker", "trainer", "transformer", "dataset")}
    ns["tracker"].transformer = ns["transformer"]
    ns["trainer"].transformer = ns["transformer"]
    ns["trainer"].dataset = ns["dataset"]
    return ns["tracker"], ns["trainer"]


def merge_args(base_args, new_args, verbose=True):
    out = copy.deepcopy(base_args)
    have = vars(out)
    for key, value in vars(new_args).items():
        if key in have:
            if value is None:
                continue
            if verbose:
                print("Overriding {} from {} to {}".format(key, have[key], value), flush=True)
        elif verbose:
            print("Setting {} to {}".format(key, value), flush=True)
        setattr(out, key, value)
    return out
