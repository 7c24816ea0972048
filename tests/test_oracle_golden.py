"""CPU: the oracle restatement against the committed golden vectors (outputs of the reference itself,
tests/golden/make_golden.py).  This is what pins the oracle."""
import glob
import os

import numpy as np
import pytest

from busca_amd import synth
from oracle import dt as odt

DT_SETS = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "dt_*.npz")))


@pytest.mark.parametrize("path", DT_SETS, ids=[os.path.basename(p)[:-4] for p in DT_SETS])
@pytest.mark.parametrize("mode", ["f64", "f32"])
def test_dt_forward_matches_reference(path, mode):
    g = np.load(path)
    d, ff, B, L, P, seed = (int(g[k]) for k in ("d", "ff", "B", "L", "P", "seed"))
    sd = synth.dt_state_dict(seed, d=d, ff=ff)
    inp = synth.dt_inputs(seed, B, L, P, sentinel_every=4 if B <= 8 else 16)
    cfg = odt.DTConfig(d=d, ff=ff, fake_f64=(mode == "f64"))
    o = odt.dt_forward(sd, cfg, **inp, return_all=True)
    # same torch ops in the same order as the reference -> expected bit-identical on the same host;
    # the tolerance only absorbs BLAS blocking differences between machines.
    np.testing.assert_allclose(o["logits"].numpy(), g["logits_" + mode], rtol=0, atol=2e-5)
    np.testing.assert_allclose(o["probs"].numpy(), g["probs_" + mode], rtol=0, atol=2e-6)
    pos = [L + 2 * j + 1 for j in range(P + 2)]
    np.testing.assert_allclose(o["hidden"][:, pos].numpy(), g["can_hidden_" + mode], rtol=0, atol=5e-5)
    np.testing.assert_allclose(o["hidden"][:, :L].mean(1).numpy(), g["mem_hidden_mean_" + mode], rtol=0, atol=5e-5)
    if "att_" + mode in g:
        att = np.stack([a.numpy() for a in o["att"]])
        np.testing.assert_allclose(att, g["att_" + mode], rtol=0, atol=2e-6)
    # chosen proposal: identical wherever the reference's top-2 margin is not a numerical tie
    ref_p = g["probs_" + mode]
    srt = np.sort(ref_p, axis=-1)
    clear = (srt[:, -1] - srt[:, -2]) > 1e-5
    assert (o["argmax"].numpy()[clear] == g["argmax_" + mode][clear]).all()


def test_activation_quirk_is_relu():
    """The reference's cloned layers run ReLU although the YAML says gelu (custom_layers.py:24-27,44-45)."""
    g = np.load(DT_SETS[0])
    d, ff, B, L, P, seed = (int(g[k]) for k in ("d", "ff", "B", "L", "P", "seed"))
    sd = synth.dt_state_dict(seed, d=d, ff=ff)
    inp = synth.dt_inputs(seed, B, L, P, sentinel_every=4)
    gelu = odt.dt_forward(sd, odt.DTConfig(d=d, ff=ff, activation="gelu"), **inp).numpy()
    assert np.abs(gelu - g["logits_f64"]).max() > 1e-3
