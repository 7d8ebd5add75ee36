"""Generates tests/golden/cdf97_pywt_small.npz with PyWavelets 1.1.1 (/opt/conda/bin/python3.9, build container only):
periodization on level inputs SHORTER than the 10-tap filter (8, 6, 4 and 2 samples), where a single fold of the linear
convolution (what pytorch_wavelets' afb1d does, oracle/cdf97.py) no longer equals the periodic transform."""
import numpy as np, pywt, warnings
warnings.simplefilter("ignore")
rng = np.random.RandomState(4242)
out = {}
for name, shape, lev in (("a", (2, 1, 64, 64), 4), ("b", (1, 1, 16, 48), 3), ("c", (1, 2, 32, 32), 5), ("d", (1, 1, 48, 24), 3)):
    x = rng.rand(*shape).astype(np.float64) - 0.5
    co = pywt.wavedec2(x, 'bior4.4', mode='periodization', level=lev, axes=(-2, -1))
    out[name + "_x"] = x
    out[name + "_levels"] = np.array(lev)
    out[name + "_ll"] = co[0]
    for i, (cH, cV, cD) in enumerate(co[1:][::-1]):      # co[-1] is the finest level
        out["%s_yh%d" % (name, i)] = np.stack((cH, cV, cD), axis=2)
    back = pywt.waverec2(co, 'bior4.4', mode='periodization', axes=(-2, -1))
    assert np.abs(back - x).max() < 1e-9
np.savez_compressed("/root/repo/tests/golden/cdf97_pywt_small.npz", **out)
print("ok", pywt.__version__)
