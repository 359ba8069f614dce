"""Host-side logic of the C ABI (no GPU): constant tables vs torch, parameter schema, error codes,
work accounting.  Runs against the CPU-emulation build of the same engine sources."""
import ctypes as C

import pytest
import torch

import helpers as Hh
from oracle import tsnet_oracle as O
from wacv23_tsnet_amd.engine import TSNetEngine


def test_linspace_matches_torch(emu_lib):
    for n in list(range(1, 70)) + [127, 128, 129, 255, 256, 512, 1000]:
        buf = (C.c_float * n)()
        emu_lib.tsnet_linspace(n, buf)
        assert torch.equal(torch.tensor(list(buf)), torch.linspace(-1, 1, n)), n


def test_coord_table_matches_reference_coord_conv(emu_lib):
    for H, W in ((8, 8), (32, 48), (256, 256)):
        buf = (C.c_float * (H * W * 3))()
        emu_lib.tsnet_coord_table(H, W, buf)
        got = torch.tensor(list(buf)).view(H, W, 3).permute(2, 0, 1)
        ref = O.coord_conv(torch.zeros(1, 0, H, W))[0]
        assert torch.equal(got[:2], ref[:2]), (H, W)            # xx, yy bit-exact
        # rr: the library's sqrtf is correctly rounded; ATen's vectorised CPU sqrt is 1 ulp low on a few
        # near-tie values (e.g. sqrt(0.50647879f)), so allow exactly that much
        assert (got[2] - ref[2]).abs().max().item() <= 1.2e-7, (H, W)
        exact = torch.sqrt((ref[0].double() ** 2 + ref[1].double() ** 2))
        assert (got[2].double() - exact).abs().max().item() <= (ref[2].double() - exact).abs().max().item() + 1e-12


@pytest.mark.parametrize("L,nb,pose", [(2, 0, False), (2, 4, False), (25, 4, True)])
def test_param_schema_is_the_checkpoint_schema(emu_lib, L, nb, pose):
    eng = TSNetEngine(label_nc=L, n_blocks=nb, pose_composite=pose, lib=emu_lib)
    assert eng.param_shapes() == O.conv_shapes(O.TSNetConfig(label_nc=L, n_blocks=nb))
    eng.close()


def test_forward_macs_match_survey_closed_form(emu_lib):
    """227.774 / 247.101 / 266.009 GMAC per frame for cfg0 / cfg2 / cfg3; cfg4 1496.695."""
    for kw, gmac in ((dict(label_nc=2, n_blocks=0), 227.774), (dict(label_nc=2, n_blocks=4), 247.101),
                     (dict(label_nc=25, n_blocks=4), 266.009), (dict(label_nc=2, n_blocks=0, n_source=5, height=512, width=512), 1496.695)):
        eng = TSNetEngine(lib=emu_lib, **kw)
        assert abs(eng.forward_macs(1) / 1e9 - gmac) < 2e-3, kw
        assert abs(eng.forward_macs(4) - 4 * eng.forward_macs(1)) < 1.0
        eng.close()


def _tiny(emu_lib, **kw):
    cfg = O.TSNetConfig(label_nc=2, n_blocks=0, n_source=1, ngf=8, enc_blocks=1, fuse_ngf=128)
    sd = O.synth_state_dict(cfg, seed=0)
    eng = TSNetEngine(label_nc=2, n_blocks=0, n_source=1, ngf=8, enc_blocks=1, height=32, width=32, max_batch=2, lib=emu_lib, **kw)
    return cfg, sd, eng


def test_create_rejects_bad_configs(emu_lib):
    for kw in (dict(n_source=0), dict(n_source=9), dict(ngf=48), dict(ngf=4), dict(height=100), dict(max_batch=0),
               dict(pose_composite=True, height=128, width=128), dict(n_downsampling=0)):
        base = dict(label_nc=2, n_blocks=0)
        base.update(kw)
        with pytest.raises(RuntimeError, match="tsnet_create failed"):
            TSNetEngine(lib=emu_lib, **base)


def test_weight_loading_errors(emu_lib):
    cfg, sd, eng = _tiny(emu_lib)
    with pytest.raises(KeyError):
        eng.load_state_dict({"img_enc.model.99.weight": torch.zeros(1)})
    eng.load_state_dict({"netD.model.0.weight": torch.zeros(3)}, strict=False)       # checkpoints carry netD too
    with pytest.raises(RuntimeError, match="shape mismatch"):
        eng.load_state_dict({"dec.map_conv.bias": torch.zeros(7)})
    some = dict(list(sd.items())[:5])
    eng.load_state_dict(some)
    with pytest.raises(RuntimeError, match="never loaded"):
        eng.finalize("cpu")
    eng.load_state_dict(sd)
    eng.finalize("cpu")
    with pytest.raises(RuntimeError, match="after finalize"):
        eng.load_state_dict(some)
    with pytest.raises(RuntimeError, match="twice"):
        eng.finalize("cpu")
    eng.close()


def test_forward_argument_errors(emu_lib):
    cfg, sd, eng = _tiny(emu_lib)
    inp = O.synth_inputs(cfg, 1, 32, 32, seed=1)
    with pytest.raises(RuntimeError, match="before finalize"):
        eng.forward(*inp)
    eng.load_state_dict(sd)
    eng.finalize("cpu")
    with pytest.raises(ValueError, match="expected shape"):
        eng.forward([inp[0][0][:, :2]], inp[1], inp[2], inp[3], inp[4])
    with pytest.raises(TypeError):
        eng.forward([inp[0][0].double()], inp[1], inp[2], inp[3], inp[4])
    big = O.synth_inputs(cfg, 3, 32, 32, seed=1)
    with pytest.raises(RuntimeError, match="max_batch"):
        eng.forward(*big)
    with pytest.raises(RuntimeError, match="set_sources"):
        eng.forward_target(inp[3], inp[4])
    with pytest.raises(RuntimeError, match="unknown stage"):
        eng.forward(*inp)
        eng.stage("nope", "cpu")
    eng.close()


def test_packed_weight_buffer_is_aliased(emu_lib):
    cfg, sd, eng = _tiny(emu_lib)
    eng.load_state_dict(sd)
    eng.finalize("cpu")
    buf = eng.packed_weights("cpu")
    assert buf.dtype == torch.uint8 and buf.numel() % 4 == 0 and buf.numel() > 4 * sum(v.numel() for v in sd.values())
    eng.close()
