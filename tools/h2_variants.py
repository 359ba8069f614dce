"""GPU box: in-run A/B of the patch-convolution variants (tools build, python -m wacv23_tsnet_amd.build --tools).

Per layer shape, every variant is timed in interleaved rounds inside ONE process (cdna_hip_programming.md rule 24) and the median
reported.  variant code of tsnet_bench_conv: tile | general kernel << 12 | ablation mask << 16 | experiment mask << 24 | XCD grid << 28.
  experiment mask (h2_tile OPT): 2 = one accumulation chain per slab, 4 = one per four slabs (the product: one per two), 8 = deep weight
    prefetch, 16 = two K groups per tile (24 = the product's single-frame tiles), 32 = weights five steps ahead, 64 = launch bounds for four
    workgroups per CU
  ablation mask  (h2_tile HABL, computes garbage): 1 no patch staging, 2 weights once, 4 A fragments once, 8 no fold, 16 no barrier
usage: h2_variants.py [rounds] [all | xcd | small | kg | chain]"""
import ctypes as C, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wacv23_tsnet_amd import _lib
lib = _lib.load_tools()
torch.zeros(1, device="cuda")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5


def code(tile=0, general=False, abl=0, opt=0, xcd=None, patch=False, w1=False, cold=False, bf16=False):
    """xcd: None = the launcher's choice, 0 = consecutive tiles per XCD, 1 / 2 / 4 / 8 = columns of the XCD grid over the N tiles"""
    gx = 0 if xcd is None else (1 if xcd == 0 else {1: 2, 2: 3, 4: 4, 8: 5}[xcd])
    return tile | (8192 if bf16 else 0) | (4096 if general else 0) | (16384 if patch else 0) | (32768 if w1 else 0) | (abl << 16) | ((opt & 15) << 24) | ((1 << 23) if opt & 16 else 0) | ((1 << 22) if opt & 32 else 0) | ((1 << 21) if cold else 0) | (gx << 28)


def run(name, shape, variants, norms=(0, 1), iters=8):
    N, H, W, Cin, Cout, k, s, p, refl = shape
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    flops = 2.0 * N * Ho * Wo * Cout * Cin * k * k
    res = {}
    for r in range(rounds):
        for vn, v in variants:
            for nrm in norms:
                ms = C.c_float()
                rc = lib.tsnet_bench_conv(N, H, W, Cin, Cout, k, s, p, refl, nrm, v, iters, C.byref(ms), None)
                res.setdefault((vn, nrm), []).append(ms.value if rc == 0 else float("nan"))
                if rc != 0 and r == 0:
                    print("   ERR", vn, lib.tsnet_op_last_error().decode())
    print(f"{name}: M={N*Ho*Wo} N={Cout} K={Cin*k*k} {flops/1e9:.1f} GFLOP")
    for (vn, nrm), t in res.items():
        med = statistics.median(t)
        print(f"   {vn:28s} {'IN+ReLU' if nrm & 1 else 'raw    '}{'+stats' if nrm & 2 else ''}  {med*1e3:8.1f} us  {flops/med/1e9:7.1f} TF   (min {min(t)*1e3:.1f} max {max(t)*1e3:.1f})", flush=True)


SEL = sys.argv[2] if len(sys.argv) > 2 else "all"
RES = (12, 32, 32, 512, 512, 3, 1, 1, 1)
if SEL == "down":       # stride-2 shapes
    for nm, shp in (("down1 (64->128)", (12, 256, 256, 64, 128, 3, 2, 1, 0)), ("down2 (128->256)", (12, 128, 128, 128, 256, 3, 2, 1, 0)),
                    ("down3 (256->512)", (12, 64, 64, 256, 512, 3, 2, 1, 0)), ("down2 B=8", (24, 128, 128, 128, 256, 3, 2, 1, 0)), ("down3 B=8", (24, 64, 64, 256, 512, 3, 2, 1, 0)),
                    ("down2 target chain (4 images)", (4, 128, 128, 128, 256, 3, 2, 1, 0)), ("down3 target chain (4 images)", (4, 64, 64, 256, 512, 3, 2, 1, 0)),
                    ("down3 one frame (3 images)", (3, 64, 64, 256, 512, 3, 2, 1, 0))):
        run(nm, shp, [("h2d 4 waves x 64", code(64, patch=True)), ("h2d 8 waves x 128", code(128, patch=True)), ("h2d 2 rows x 128", code(2128, patch=True)),
                      ("general 128", code(128, general=True)), ("the layer's own", code(0))], norms=(1,))
    sys.exit(0)
if SEL == "bf16":       # the bf16-operand kernels against the fp16 x 2 ones, layer by layer at the configs[2] batch (24 source images)
    for nm, shp in (("stem (8->64, 7x7)", (24, 256, 256, 8, 64, 7, 1, 3, 1)), ("down1 (64->128)", (24, 256, 256, 64, 128, 3, 2, 1, 0)),
                    ("down2 (128->256)", (24, 128, 128, 128, 256, 3, 2, 1, 0)), ("down3 (256->512)", (24, 64, 64, 256, 512, 3, 2, 1, 0)),
                    ("res (512->512)", (24, 32, 32, 512, 512, 3, 1, 1, 1)), ("dec_up1 (256->128 @128^2)", (8, 128, 128, 256, 128, 3, 1, 1, 1)),
                    ("dec_up2 (128->64 @256^2)", (8, 256, 256, 128, 64, 3, 1, 1, 1))):
        vs = [("fp16x2, the layer's own", code(0)), ("bf16, the layer's own", code(0, bf16=True))]
        if shp[5] == 3 and shp[6] == 1:
            vs += [("bf16 w1 (winograd-x)", code(w1=True, bf16=True)), ("bf16 4x64", code(64, bf16=True)), ("bf16 2x128", code(2128, bf16=True))]
            if shp[4] % 128 == 0:
                vs += [("bf16 4x128 (2x2 waves)", code(128, bf16=True)), ("bf16 4x128 (1x4 waves)", code(3128, bf16=True))]
                vs += [(f"bf16 4x128 1x4 abl{m}", code(3128, bf16=True, abl=m)) for m in (1, 2, 4, 8, 3, 7, 16)]
                vs += [(f"bf16 4x128 abl{m}", code(128, bf16=True, abl=m)) for m in (1, 2, 4, 3, 7, 16)]
        if shp[6] == 2 and shp[3] >= 128:
            vs += [("bf16 h2d 4 waves x 64", code(64, patch=True, bf16=True)), ("bf16 h2d 8 waves x 128", code(128, patch=True, bf16=True)), ("bf16 general 128", code(128, general=True, bf16=True))]
        run(nm, shp, vs, norms=(0,) if shp[5] == 7 else (1,))
    sys.exit(0)
if SEL == "downsmall":  # stride-2 shapes of a single frame
    for nm, shp in (("down2 one frame (3 images)", (3, 128, 128, 128, 256, 3, 2, 1, 0)), ("down3 one frame (3 images)", (3, 64, 64, 256, 512, 3, 2, 1, 0)),
                    ("down2 1 image", (1, 128, 128, 128, 256, 3, 2, 1, 0)), ("down3 1 image", (1, 64, 64, 256, 512, 3, 2, 1, 0)),
                    ("down2 2 frames (6 images)", (6, 128, 128, 128, 256, 3, 2, 1, 0)), ("down3 2 frames (6 images)", (6, 64, 64, 256, 512, 3, 2, 1, 0))):
        run(nm, shp, [("h2d 4 waves x 64", code(64, patch=True)), ("h2d 8 waves x 128", code(128, patch=True)), ("h2d 2 rows x 128", code(2128, patch=True)),
                      ("the layer's own", code(0))], norms=(1,))
    sys.exit(0)
if SEL == "w1":         # Winograd F(2,3) along x (conv_w1.hpp) against the direct patch kernel, layer by layer
    W1 = ("w1 (winograd-x)", code(w1=True))
    run("res (B=4: 12 images)", RES, [("4x64 direct", code(64)), W1, ("4x64 direct, cold weights", code(64, cold=True)), ("w1, cold weights", code(w1=True, cold=True))], iters=24)
    run("res w1 ablations", RES, [W1] + [(f"w1 abl{m}", code(w1=True, abl=m)) for m in (1, 2, 4, 3, 7, 8, 15, 16, 17)], norms=(0,))
    run("res (B=8: 24 images)", (24, 32, 32, 512, 512, 3, 1, 1, 1), [("4x64 direct", code(64)), W1])
    run("res (B=1: 3 images)", (3, 32, 32, 512, 512, 3, 1, 1, 1), [("4x64 deep kg2", code(64, opt=24)), W1])
    run("res clip (1 image)", (1, 32, 32, 512, 512, 3, 1, 1, 1), [("4x32 deep kg2", code(32, opt=24)), W1])
    run("fuse_c2 (1024->1024)", (12, 32, 32, 1024, 1024, 3, 1, 1, 1), [("4x128 direct", code(128)), W1, ("w1 grid x2", code(w1=True, xcd=2)), ("w1 grid x4", code(w1=True, xcd=4))])
    run("fuse_c1_src (512->1024)", (12, 32, 32, 512, 1024, 3, 1, 1, 1), [("4x128 direct", code(128)), W1], norms=(0,))
    run("fuse_c1_tar (512->1024, 4 images)", (4, 32, 32, 512, 1024, 3, 1, 1, 1), [("the layer's own", code(0)), W1], norms=(0,))
    run("dec_up0 (512->256 @64^2)", (4, 64, 64, 512, 256, 3, 1, 1, 1), [("4x128 direct", code(128)), W1], norms=(0,))
    run("dec_up1 (256->128 @128^2)", (4, 128, 128, 256, 128, 3, 1, 1, 1), [("the layer's own", code(0)), W1], norms=(0,))
    run("dec_up2 (128->64 @256^2)", (4, 256, 256, 128, 64, 3, 1, 1, 1), [("the layer's own", code(0)), W1], norms=(0,))
    sys.exit(0)
if SEL == "w1abl":      # conv_w1 on the ResnetBlock layer (raw input), sustained loop: the product and the ablations (garbage): 1 no producer work, 2 weights once, 3 both
    W1 = ("w1 (product)", code(w1=True))
    run("res w1 ablations", RES, [W1] + [(f"w1 abl{m}", code(w1=True, abl=m)) for m in (1, 2, 3)], norms=(0,), iters=24)
    sys.exit(0)
if SEL == "w1c":        # conv_w1: tiles per workgroup (tile code 1, 2, 3; 0 = the launcher's choice), with the output statistics (norm bit 1) as in the forward
    for nm, shp in (("res (B=4: 12 images)", RES), ("res (B=8: 24 images)", (24, 32, 32, 512, 512, 3, 1, 1, 1)), ("fuse_c2 (1024->1024)", (12, 32, 32, 1024, 1024, 3, 1, 1, 1)),
                    ("fuse_c1_src (512->1024)", (12, 32, 32, 512, 1024, 3, 1, 1, 1)), ("fuse_c1_tar (4 images)", (4, 32, 32, 512, 1024, 3, 1, 1, 1)),
                    ("dec_up0 (512->256 @64^2)", (4, 64, 64, 512, 256, 3, 1, 1, 1)), ("dec_res (B=8: 8 images)", (8, 32, 32, 512, 512, 3, 1, 1, 1))):
        run(nm, shp, [(f"w1 chunk {c}" if c else "w1 launcher's choice", code(c, w1=True)) for c in (1, 2, 3, 0)], norms=(2, 3), iters=18)
    sys.exit(0)
if SEL == "w1dec":      # the decoder's second and third up-convolutions: direct kernel (the forward's choice in round 4) against conv_w1, B = 4 and one frame
    for nm, shp in (("dec_up1 (256->128 @128^2) B=4", (4, 128, 128, 256, 128, 3, 1, 1, 1)), ("dec_up2 (128->64 @256^2) B=4", (4, 256, 256, 128, 64, 3, 1, 1, 1)),
                    ("dec_up1 B=1", (1, 128, 128, 256, 128, 3, 1, 1, 1)), ("dec_up2 B=1", (1, 256, 256, 128, 64, 3, 1, 1, 1)),
                    ("dec_up1 B=8", (8, 128, 128, 256, 128, 3, 1, 1, 1)), ("dec_up2 B=8", (8, 256, 256, 128, 64, 3, 1, 1, 1))):
        vs = [("direct, the launcher's tile", code(0, patch=True)), ("w1", code(0, w1=True)), ("w1 chunk 1", code(1, w1=True)), ("w1 chunk 2", code(2, w1=True))]
        if shp[0] == 1:
            vs.append(("direct 4x64 deep kg2 (B=1 form)", code(64, opt=24)))
        run(nm, shp, vs, norms=(2,), iters=12)
    sys.exit(0)
if SEL == "w1x":        # conv_w1 on the ResnetBlock layer: XCD grids over the tile matrix, warm and cold weights (a fresh copy of the planes per launch)
    for cold in (False, True):
        run("res (12 images)" + (", cold weights" if cold else ""), RES, [("w1 linear", code(0, w1=True, xcd=0, cold=cold))] + [(f"w1 grid x{g}", code(0, w1=True, xcd=g, cold=cold)) for g in (2, 4, 8)], norms=(2, 3), iters=24)
    run("fuse_c1_src (512->1024)", (12, 32, 32, 512, 1024, 3, 1, 1, 1), [("w1 launcher", code(0, w1=True))] + [(f"w1 grid x{g}", code(0, w1=True, xcd=g)) for g in (0, 2, 4, 8)], norms=(2,), iters=12)
    sys.exit(0)
if SEL == "chain":      # slabs per accumulation chain: 1, 2 (product), 4; and the ablations of the 4 x 64 tile
    run("res", RES, [("4x64 chain 2 (product)", code(64)), ("4x64 chain 1", code(64, opt=2)), ("4x64 chain 4", code(64, opt=4)),
                     ("4x64 weights 5 ahead", code(64, opt=32)), ("4x64 chain 4 + 5 ahead", code(64, opt=36)), ("2x128 weights 5 ahead", code(2128, opt=32)),
                     ("4x128 chain 2 (product)", code(128)), ("4x128 chain 1", code(128, opt=2)), ("4x128 chain 4", code(128, opt=4)), ("2x128", code(2128))])
    run("res ablations 4x64", RES, [("full", code(64))] + [(f"abl{m}", code(64, abl=m)) for m in (1, 2, 4, 7, 8, 16)], norms=(0,))
    run("fuse_c2", (12, 32, 32, 1024, 1024, 3, 1, 1, 1), [("4x128 chain 2 (product)", code(128)), ("4x128 chain 1", code(128, opt=2)), ("4x128 chain 4", code(128, opt=4)), ("4x64", code(64))])
    run("fuse_c1_src (512->1024)", (12, 32, 32, 512, 1024, 3, 1, 1, 1), [("4x64", code(64)), ("4x128", code(128))], norms=(0,))
    run("dec_up0 (512->256 @64^2)", (4, 64, 64, 512, 256, 3, 1, 1, 1), [("4x64", code(64)), ("4x128", code(128))], norms=(0,))
    sys.exit(0)
if SEL == "kg":         # two K groups per tile (eight waves, one workgroup per CU) against the co-resident four-wave workgroups
    for nm, n in (("res B=4 (12 images)", 12), ("res B=2 (6 images)", 6), ("res B=1 (3 images)", 3), ("res clip (1 image)", 1), ("res B=8 (24 images)", 24)):
        run(nm, (n, 32, 32, 512, 512, 3, 1, 1, 1), [("4x64", code(64)), ("4x64 deep", code(64, opt=8)), ("4x64 deep kg2", code(64, opt=24)), ("4x32 deep kg2", code(32, opt=24)), ("4x64 kg2", code(64, opt=16))])
    run("fuse_c2 (12 images)", (12, 32, 32, 1024, 1024, 3, 1, 1, 1), [("4x128", code(128)), ("4x64 deep kg2", code(64, opt=24)), ("4x128 kg2", code(128, opt=16))])
    run("fuse_c2 (3 images)", (3, 32, 32, 1024, 1024, 3, 1, 1, 1), [("4x128", code(128)), ("4x64 deep kg2", code(64, opt=24)), ("4x128 kg2", code(128, opt=16))])
    run("fuse_c1_src (12 images)", (12, 32, 32, 512, 1024, 3, 1, 1, 1), [("4x128", code(128)), ("4x64 deep kg2", code(64, opt=24))], norms=(0,))
    run("dec_up0 (B=4)", (4, 64, 64, 512, 256, 3, 1, 1, 1), [("4x128", code(128)), ("4x64", code(64)), ("4x64 deep kg2", code(64, opt=24))], norms=(0,))
    run("dec_up0 (B=1)", (1, 64, 64, 512, 256, 3, 1, 1, 1), [("4x64 deep", code(64, opt=8)), ("4x64 deep kg2", code(64, opt=24)), ("4x32 deep kg2", code(32, opt=24))], norms=(0,))
    run("dec_up1 (B=4)", (4, 128, 128, 256, 128, 3, 1, 1, 1), [("4x128", code(128)), ("4x64", code(64)), ("4x64 deep kg2", code(64, opt=24))], norms=(0,))
    run("dec_up1 (B=1)", (1, 128, 128, 256, 128, 3, 1, 1, 1), [("4x64 deep", code(64, opt=8)), ("4x64 deep kg2", code(64, opt=24))], norms=(0,))
    sys.exit(0)
if SEL == "small":      # one driving frame: launches of at most one workgroup per CU
    for nm, n in (("res B=1 (3 images)", 3), ("res clip (1 image)", 1), ("res B=2 (6 images)", 6)):
        run(nm, (n, 32, 32, 512, 512, 3, 1, 1, 1), [("4x32", code(32)), ("4x64", code(64)), ("4x32 deep", code(32, opt=8)), ("4x64 deep", code(64, opt=8)), ("2x128", code(2128))])
    run("fuse_c2 B=1 (3 images)", (3, 32, 32, 1024, 1024, 3, 1, 1, 1), [("4x128", code(128)), ("4x64", code(64)), ("4x64 deep", code(64, opt=8)), ("4x32 deep", code(32, opt=8))])
    run("fuse_c1_tar B=1 (1 image)", (1, 32, 32, 512, 1024, 3, 1, 1, 1), [("4x128", code(128)), ("4x64", code(64)), ("4x64 deep", code(64, opt=8)), ("4x32 deep", code(32, opt=8))], norms=(0,))
    run("dec_up0 B=1", (1, 64, 64, 512, 256, 3, 1, 1, 1), [("4x32", code(32)), ("4x64", code(64)), ("4x32 deep", code(32, opt=8)), ("4x64 deep", code(64, opt=8))], norms=(0,))
    run("dec_up1 B=1", (1, 128, 128, 256, 128, 3, 1, 1, 1), [("4x64", code(64)), ("4x32 deep", code(32, opt=8)), ("4x64 deep", code(64, opt=8))], norms=(0,))
    run("dec_up2 B=1", (1, 256, 256, 128, 64, 3, 1, 1, 1), [("4x64", code(64)), ("4x32", code(32)), ("4x64 deep", code(64, opt=8))], norms=(0,))
    for nm, shp in (("down1 B=1 src", (3, 256, 256, 64, 128, 3, 2, 1, 0)), ("down2 B=1 src", (3, 128, 128, 128, 256, 3, 2, 1, 0)), ("down3 B=1 src", (3, 64, 64, 256, 512, 3, 2, 1, 0)),
                    ("down2 1 image", (1, 128, 128, 128, 256, 3, 2, 1, 0)), ("down3 1 image", (1, 64, 64, 256, 512, 3, 2, 1, 0))):
        run(nm, shp, [("h2d 4 waves x 64", code(64)), ("h2d 8 waves x 128", code(128)), ("general 64", code(64, general=True)), ("general 128", code(128, general=True))], norms=(1,))
    run("1x1 (1024->512) 1 image", (1, 32, 32, 1024, 512, 1, 1, 0, 0), [("general 64", code(64))], norms=(0,))
    sys.exit(0)
FUSE = (12, 32, 32, 1024, 1024, 3, 1, 1, 1)
if SEL in ("all", "xcd"):
    run("res: XCD grids (4x64)", RES, [("linear", code(64, xcd=0))] + [(f"grid x{g}", code(64, xcd=g)) for g in (1, 2, 4, 8)])
    run("res: XCD grids (2x128)", RES, [("linear", code(2128, xcd=0))] + [(f"grid x{g}", code(2128, xcd=g)) for g in (1, 2, 4)], norms=(1,))
    run("fuse_c2: XCD grids (4x128)", FUSE, [("linear", code(128, xcd=0))] + [(f"grid x{g}", code(128, xcd=g)) for g in (1, 2, 4, 8)])
    run("fuse_c2: XCD grids (4x64)", FUSE, [("linear", code(64, xcd=0))] + [(f"grid x{g}", code(64, xcd=g)) for g in (2, 4, 8)], norms=(1,))
    run("fuse_c1_src: XCD grids (4x128)", (12, 32, 32, 512, 1024, 3, 1, 1, 1), [("linear", code(128, xcd=0))] + [(f"grid x{g}", code(128, xcd=g)) for g in (2, 4, 8)], norms=(0,))
    run("dec_up0: XCD grids (4x128)", (4, 64, 64, 512, 256, 3, 1, 1, 1), [("linear", code(128, xcd=0))] + [(f"grid x{g}", code(128, xcd=g)) for g in (1, 2)], norms=(0,))
    for nm, shp in (("down1 (64->128)", (12, 256, 256, 64, 128, 3, 2, 1, 0)), ("down2 (128->256)", (12, 128, 128, 128, 256, 3, 2, 1, 0)),
                    ("down3 (256->512)", (12, 64, 64, 256, 512, 3, 2, 1, 0))):
        run(nm, shp, [("h2d 4 waves x 64", code(64)), ("h2d 8 waves x 128", code(128)), ("general 128", code(128, general=True))], norms=(1,))
if SEL == "xcd":
    sys.exit(0)
run("res (ResnetBlock conv, B=4 K=3)", RES,
    [("4x64", code(64)), ("4x64 chain 1", code(64, opt=2)), ("4x64 chain 4", code(64, opt=4)),
     ("4x128", code(128)), ("2x128", code(2128)), ("4x32", code(32))])
run("res ablations 4x64", RES,
    [("full", code(64))] + [(f"abl{m}", code(64, abl=m)) for m in (1, 2, 4, 7, 8, 16, 15, 31)], norms=(0,))
run("res ablations 2x128", RES, [("full", code(2128))] + [(f"abl{m}", code(2128, abl=m)) for m in (1, 2, 7)], norms=(0,))
run("res B=1 (3 images)", (3, 32, 32, 512, 512, 3, 1, 1, 1), [("4x32", code(32)), ("4x64", code(64)), ("2x128", code(2128))])
run("fuse_c2 (1024->1024)", (12, 32, 32, 1024, 1024, 3, 1, 1, 1),
    [("4x64", code(64)), ("4x128", code(128)), ("2x128", code(2128))])
run("fuse_c1_src (512->1024)", (12, 32, 32, 512, 1024, 3, 1, 1, 1), [("4x64", code(64)), ("4x128", code(128)), ("2x128", code(2128))], norms=(0,))
run("dec_up0 (512->256 @64^2)", (4, 64, 64, 512, 256, 3, 1, 1, 1), [("4x64", code(64)), ("4x128", code(128)), ("2x128", code(2128))], norms=(0,))
run("dec_up1 (256->128 @128^2)", (4, 128, 128, 256, 128, 3, 1, 1, 1), [("4x64", code(64)), ("4x128", code(128)), ("2x128", code(2128))], norms=(0,))
run("dec_up2 (128->64 @256^2)", (4, 256, 256, 128, 64, 3, 1, 1, 1), [("4x32", code(32)), ("4x64", code(64))], norms=(0,))
for nm, shp in (("down1 (64->128)", (12, 256, 256, 64, 128, 3, 2, 1, 0)), ("down2 (128->256)", (12, 128, 128, 128, 256, 3, 2, 1, 0)),
                ("down3 (256->512)", (12, 64, 64, 256, 512, 3, 2, 1, 0))):
    run(nm, shp, [("h2d 4 waves x 64", code(64)), ("h2d 8 waves x 128", code(128)), ("general 64", code(64, general=True)),
                  ("general 128", code(128, general=True))], norms=(1,))
run("stem (8->64, 7x7)", (12, 256, 256, 8, 64, 7, 1, 3, 1), [("h2s", code(0)), ("general", code(64, general=True))], norms=(0,))
run("stem pose (32->64, 7x7)", (12, 256, 256, 32, 64, 7, 1, 3, 1), [("general", code(0))], norms=(0,))
run("1x1 (1024->512)", (4, 32, 32, 1024, 512, 1, 1, 0, 0), [("general 64", code(64))], norms=(0,))
