import ctypes, sys, os, shutil
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
P = os.path.join(R, "imagecompressionlearnedliftingandlearnedtreebasedmodels_amd")
shutil.copy(os.path.join(P, "liblldwt.so"), "/tmp/keep.so")
shutil.copy(os.path.join(R, "ab_old", "dbg.so"), os.path.join(P, "liblldwt.so"))
try:
    from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd import ops, _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    Pn, B, h, w, C_, K = 3, 8, int(os.environ.get("DH", 256)), int(os.environ.get("DW", 512)), 16, 5
    Z = Pn * B
    g = torch.Generator(device=dev).manual_seed(1)
    ws = [torch.randn(Pn, 16, 1, K, K, device=dev) * .1, torch.zeros(Pn, 16, device=dev), torch.randn(Pn, 16, 16, K, K, device=dev) * .05,
          torch.zeros(Pn, 16, device=dev), torch.randn(Pn, 16, 16, K, K, device=dev) * .05, torch.zeros(Pn, 16, device=dev),
          torch.randn(Pn, 1, 16, K, K, device=dev) * .1, torch.zeros(Pn, 1, device=dev)]
    packed = ops.pack_pblock(*ws)
    taps = torch.tensor([[0., -1.5, -1.5]] * Pn, device=dev)
    src = torch.randn(Z, h, w, device=dev); dst = torch.randn(Z, h, w, device=dev); out = torch.empty_like(dst)
    for _ in range(3):
        ops.lift_step(ops.view_of(src, Z, h, w), ops.view_of(dst, Z, h, w), ops.view_of(out, Z, h, w), Z, B, h, w, taps, packed, C_, K, True, 1.0, 0.1)
    torch.cuda.synchronize()
    buf = np.zeros(64 * 16, dtype=np.uint64)
    lib.lldwt_debug_read.argtypes = [ctypes.c_void_p]
    print("rc", lib.lldwt_debug_read(buf.ctypes.data))
    t = buf.reshape(64, 16).astype(np.float64)
    for wg in (0, 1, 17, 63):
        d = (t[wg] - t[wg, 0]) / 100.0     # us at 100 MHz
        print(wg, np.round(d, 2))
finally:
    shutil.copy("/tmp/keep.so", os.path.join(P, "liblldwt.so"))
