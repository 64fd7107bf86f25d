"""Standalone FourierBasicBlock / CondFourierBasicBlock: the oracle (CPU)
and the HIP op (GPU) against outputs of the REAL reference blocks (tests/golden/ops_fourier.npz)."""
import json
import os

import numpy as np
import pytest

from helpers import GOLDEN, rel_l2


def _load():
    d = np.load(os.path.join(GOLDEN, "ops_fourier.npz"))
    return json.loads(bytes(d["meta"]).decode()), d


def _weights(keys, seed):
    from lns_amd import filler
    sd = {}
    for ks in keys:
        k, shp = str(ks).split(":")
        sd[k] = filler.fill_tensor(k, tuple(int(v) for v in shp.split("x")), seed)
    return sd


def _inputs(m):
    from lns_amd import filler
    x = filler.normal("xop", (m["B"], m["C"], m["H"], m["W"]), m["input_seed"])
    cond = filler.normal("cop", (m["B"], m["C"]), m["input_seed"])
    return x, cond


def test_oracle_fourier_blocks_match_reference():
    import lns_oracle
    m, d = _load()
    x, cond = _inputs(m)
    sd = {"blk." + k: v for k, v in _weights(d["fourier_keys"], m["weight_seed"]).items()}
    net = lns_oracle._Net(sd, (0, 0))
    assert rel_l2(lns_oracle.fourier_basic_block(net, x, "blk"), d["fourier_y"]) < 2e-6
    sd = {"blk." + k: v for k, v in _weights(d["cond_fourier_keys"], m["weight_seed"]).items()}
    assert rel_l2(lns_oracle.cond_fourier_basic_block(sd, "blk", x, cond), d["cond_fourier_y"]) < 2e-6


@pytest.mark.gpu
@pytest.mark.parametrize("conditional", [False, True])
def test_hip_fourier_blocks_match_reference(conditional):
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from lns_amd.modules.fourier_cond import CondFourierBasicBlock, FourierBasicBlock
    m, d = _load()
    x, cond = _inputs(m)
    name = "cond_fourier" if conditional else "fourier"
    blk = (CondFourierBasicBlock if conditional else FourierBasicBlock)(m["C"], m["C"], [m["m1"], m["m2"]])
    w = _weights(d[name + "_keys"], m["weight_seed"])
    assert set(blk.state_dict()) == set(w)
    blk.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=True)
    xd = torch.from_numpy(x).cuda()
    y = blk(xd, torch.from_numpy(cond).cuda()) if conditional else blk(xd)
    y = y.cpu().numpy()
    assert rel_l2(y, d[name + "_y"]) < 5e-6
    assert rel_l2(y, d[name + "_y_f64"]) < 5e-6
    # the block object keeps its weights on the device: a second forward uploads nothing and gives the same bits, another
    # batch size re-plans, and a weight written in place is seen (the handle is rebuilt from the parameters' versions)
    cd = torch.from_numpy(cond).cuda()
    run = (lambda xx, cc: blk(xx, cc)) if conditional else (lambda xx, cc: blk(xx))
    h0 = blk._h.value
    assert np.array_equal(run(xd, cd).cpu().numpy(), y) and blk._h.value == h0
    assert np.array_equal(run(xd[:1].contiguous(), cd[:1].contiguous()).cpu().numpy(), y[:1])
    assert np.array_equal(run(xd, cd).cpu().numpy(), y)
    with torch.no_grad():
        blk.conv.bias.add_(1.0)
    y3 = run(xd, cd).cpu().numpy()
    assert not np.array_equal(y3, y) and np.isfinite(y3).all()


def _gen_cases():
    m, _ = _load()
    return [g["name"] for g in m.get("general", [])]


def _gen(m, name):
    from lns_amd import filler
    g = [e for e in m["general"] if e["name"] == name][0]
    x = filler.normal("xop_" + name, (m["B"], g["cin"], m["H"], m["W"]), m["input_seed"])
    cond = filler.normal("cop_" + name, (m["B"], g["cin"]), m["input_seed"])
    return g, x, cond


@pytest.mark.parametrize("name", _gen_cases())
def test_oracle_general_fourier_blocks_match_reference(name):
    """in_planes != planes, residual=False and the other ACTIVATION_REGISTRY entries (modules/basics.py:531-583)."""
    import lns_oracle
    m, d = _load()
    g, x, cond = _gen(m, name)
    sd = {"blk." + k: v for k, v in _weights(d[name + "_keys"], m["weight_seed"]).items()}
    if g["cond"]:
        y = lns_oracle.cond_fourier_basic_block(sd, "blk", x, cond, residual=g["residual"])
    else:
        y = lns_oracle.fourier_basic_block(lns_oracle._Net(sd, (0, 0)), x, "blk", act=g["act"], residual=g["residual"])
    assert y.shape == d[name + "_y"].shape
    assert rel_l2(y, d[name + "_y"]) < 2e-6


@pytest.mark.gpu
@pytest.mark.parametrize("name", _gen_cases())
def test_hip_general_fourier_blocks_match_reference(name):
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from lns_amd.modules.fourier_cond import CondFourierBasicBlock, FourierBasicBlock
    m, d = _load()
    g, x, cond = _gen(m, name)
    if g["cond"]:
        blk = CondFourierBasicBlock(g["cin"], g["cout"], [m["m1"], m["m2"]], residual=g["residual"])
    else:
        blk = FourierBasicBlock(g["cin"], g["cout"], [m["m1"], m["m2"]], activation=g["act"], residual=g["residual"])
    w = _weights(d[name + "_keys"], m["weight_seed"])
    assert set(blk.state_dict()) == set(w)
    blk.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=True)
    xd = torch.from_numpy(x).cuda()
    y = blk(xd, torch.from_numpy(cond).cuda()) if g["cond"] else blk(xd)
    y = y.cpu().numpy()
    assert y.shape == d[name + "_y"].shape
    assert rel_l2(y, d[name + "_y"]) < 5e-6
    assert rel_l2(y, d[name + "_y_f64"]) < 5e-6


def test_residual_needs_matching_planes():
    pytest.importorskip("torch")
    from lns_amd.modules.fourier_cond import FourierBasicBlock
    with pytest.raises(ValueError):
        FourierBasicBlock(4, 6, [2, 2], residual=True)
    with pytest.raises(NotImplementedError):
        FourierBasicBlock(4, 4, [2, 2], activation="softplus")
