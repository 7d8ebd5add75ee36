"""Generates tests/golden/cdf97_pywt.npz with PyWavelets 1.1.1: run with /opt/conda/bin/python3.9 (build container only)."""
import numpy as np, pywt
rng = np.random.RandomState(1337)
x = rng.rand(2, 1, 32, 48).astype(np.float64)
out = {"x": x}
co = pywt.wavedec2(x, 'bior4.4', mode='periodization', level=2, axes=(-2,-1))
out["ll"] = co[0]
# co[1] is coarsest detail (cH,cV,cD); co[-1] finest
for i, (cH, cV, cD) in enumerate(co[1:][::-1]):
    out["lh%d" % i] = cH; out["hl%d" % i] = cV; out["hh%d" % i] = cD
w = pywt.Wavelet('bior4.4')
out["dec_lo"] = np.array(w.dec_lo); out["dec_hi"] = np.array(w.dec_hi); out["rec_lo"] = np.array(w.rec_lo); out["rec_hi"] = np.array(w.rec_hi)
np.savez("/root/repo/tests/golden/cdf97_pywt.npz", **out)
print("ok", pywt.__version__)
