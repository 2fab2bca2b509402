"""Block-level forward + backward against fixtures produced by the REFERENCE's own block classes (conv_block, up_conv,
Attention_block: next_best_path/networks/nbp_model.py:8-62) in train mode, evaluated in float64 by tests/golden/make_golden.py
(gen_blocks).  The cases (nextbestpath_amd/utility/synthetic.py::BLOCK_CASES) keep every ReLU pre-activation clear of zero, so the
blocks are smooth and two fp32 arithmetics agree to rounding: the HIP autograd.Functions are held at 1e-5 of each tensor's
maximum (stock torch CPU fp32 sits at 1e-7 .. 4e-6 on the same tensors: recorded in the fixture), the oracle restatement at 1e-9
in float64 and 1e-5 in float32.  This is the reference-held evidence for SURVEY 8 row A3 (the whole-network gradient goldens
nbp_train_*.npz are chaotic at fp32 and only hold 3e-2 / 15 %)."""
import os

import numpy as np
import pytest
import torch

from nextbestpath_amd.utility.synthetic import BLOCK_CASES, make_block_case

TAGS = [r[0] for r in BLOCK_CASES]


def _golden(golden_dir):
    return np.load(os.path.join(golden_dir, "nbp_blocks_bwd.npz"))


def _pin(g, tag, sd, inputs, dy):
    want = g[f"{tag}__pin"]
    got = [float(sum(x.double().sum() for x in inputs)), float(dy.double().sum()),
           float(sum(v.double().sum() for v in sd.values() if v.is_floating_point()))]
    assert np.allclose(got, want, rtol=0, atol=1e-6), f"{tag}: the seeded case drifted from the one the fixture was made on"


def _check(g, tag, name, got, tol, zero_scale=None):
    """got: a tensor (any float dtype / device) in the reference's layout; compared on the fixture's strided samples + its sum."""
    want, st = g[f"{tag}__{name}"], g[f"{tag}__{name}__stats"]
    stride, s_all, s_abs, mx = int(st[0]), st[1], st[2], st[3]
    f = got.detach().double().cpu().flatten()
    smp = f[::stride].numpy()
    assert smp.shape == want.shape, (tag, name, smp.shape, want.shape)
    if mx < 1e-9:            # analytically zero (a bias in front of a batch-statistics BatchNorm): rounding noise on both sides
        assert zero_scale is not None and float(f.abs().max()) <= 1e-5 * zero_scale, (tag, name, float(f.abs().max()), zero_scale)
        return 0.0
    err = float(np.abs(smp - want).max()) / mx
    assert err <= tol, (tag, name, err, tol)
    assert abs(float(f.sum()) - s_all) <= tol * max(s_abs, mx), (tag, name, "sum")        # every entry, not only the samples
    return err


def _names(kind):
    return {"conv_block": ["conv.0", "conv.1", "conv.3", "conv.4"], "up_conv": ["up.1", "up.2"],
            "attention": ["W_g.0", "W_g.1", "W_x.0", "W_x.1", "psi.0", "psi.1"]}[kind]


@pytest.mark.parametrize("tag", TAGS)
@pytest.mark.parametrize("dt,tol", [(torch.float64, 1e-9), (torch.float32, 1e-5)])
def test_oracle_blocks_vs_reference_golden(golden_dir, tag, dt, tol):
    """oracle/nbp_net.py's conv_block / up_conv / attention in train mode, forward and autograd backward."""
    from oracle import nbp_net
    g = _golden(golden_dir)
    kind, cin, cout, sd, inputs, dy = make_block_case(tag)
    _pin(g, tag, sd, inputs, dy)
    torch.set_num_threads(8)
    sdp = {"B." + k: (v.to(dt).clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    xs = [x.to(dt).clone().requires_grad_(True) for x in inputs]
    if kind == "conv_block":
        y = nbp_net.conv_block(sdp, "B", torch.cat(xs, 1) if len(xs) > 1 else xs[0], train=True)
    elif kind == "up_conv":
        y = nbp_net.up_conv(sdp, "B", xs[0], train=True)
    else:
        y = nbp_net.attention(sdp, "B", xs[0], xs[1], train=True)
    y.backward(dy.to(dt))
    _check(g, tag, "y", y, tol)
    for i, x in enumerate(xs):
        _check(g, tag, f"dx{i}", x.grad, tol)
    for lay in _names(kind):
        wkey = f"B.{lay}.weight"
        zs = float(sdp[wkey].grad.abs().max())
        for leaf in ("weight", "bias"):
            _check(g, tag, "d__" + f"{lay}.{leaf}".replace(".", "__"), sdp[f"B.{lay}.{leaf}"].grad, tol, zero_scale=zs)


def _module(kind, cin, cout, sd, dev):
    """The product's own parameter containers (networks/nbp_model.py: reference key names), loaded strictly."""
    from nextbestpath_amd.networks import nbp_model as nm
    m = {"conv_block": lambda: nm._double_conv(sum(cin), cout), "up_conv": lambda: nm._up_conv(cin[0], cout),
         "attention": lambda: nm._Gate(cin[0], cin[1], cout)}[kind]()
    m.load_state_dict(sd, strict=True)
    return m.to(dev).train()


@pytest.mark.gpu
@pytest.mark.parametrize("gate_fuse", [True, False])
@pytest.mark.parametrize("tag", TAGS)
def test_hip_blocks_vs_reference_golden(hip, golden_dir, tag, gate_fuse, monkeypatch):
    """networks/training.py::_block / _up_conv / _gate (the per-layer autograd.Functions over csrc/nbp_train.hip, nbp_split.hip)
    against the reference block's float64 forward / backward: outputs, input gradients, every parameter gradient and the
    BatchNorm running statistics after the forward, all within 1e-5 of the tensor's maximum.  gate_fuse: the attention gate's
    element-wise middle as one fused Function (round 6, the default) or as the separate Functions of rounds 3-5."""
    from nextbestpath_amd.networks import training as tr
    if not gate_fuse and not tag.startswith("att"):
        pytest.skip("gate_fuse only changes Attention_block")
    monkeypatch.setattr(tr, "_GATE_FUSE", gate_fuse)
    g = _golden(golden_dir)
    kind, cin, cout, sd, inputs, dy = make_block_case(tag)
    _pin(g, tag, sd, inputs, dy)
    dev = torch.device("cuda")
    m = _module(kind, cin, cout, sd, dev)
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(dev)
    nchw = lambda t: t.permute(0, 3, 1, 2)
    xs = [nhwc(x).requires_grad_(True) for x in inputs]
    tr._reset_arena(dev)
    if kind == "conv_block":
        y = tr._block(m.conv, xs[0], xs[1] if len(xs) > 1 else None)
    elif kind == "up_conv":
        y = tr._up_conv(m.up, xs[0])
    else:
        y = tr._gate(m, xs[0], xs[1])
    y.backward(nhwc(dy))
    torch.cuda.synchronize()
    tol = 1e-5
    worst = {"y": _check(g, tag, "y", nchw(y), tol)}
    for i, x in enumerate(xs):
        worst[f"dx{i}"] = _check(g, tag, f"dx{i}", nchw(x.grad), tol)
    named = dict(m.named_parameters())
    for lay in _names(kind):
        zs = float(named[f"{lay}.weight"].grad.abs().max())
        for leaf in ("weight", "bias"):
            k = f"{lay}.{leaf}"
            worst[k] = _check(g, tag, "d__" + k.replace(".", "__"), named[k].grad, tol, zero_scale=zs)
    for k, b in m.named_buffers():
        if k.endswith("running_mean") or k.endswith("running_var"):
            worst[k] = _check(g, tag, "buf__" + k.replace(".", "__"), b, tol)
    print(tag, "worst error / max:", {k: f"{v:.1e}" for k, v in worst.items()})
