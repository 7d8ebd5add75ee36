"""CPU oracle for the learned-lifting DWT + CNN entropy-model hot path.

TEST INFRASTRUCTURE ONLY.  This package is a CPU restatement (torch CPU ops /
numpy, fp32) of the reference algorithm, every function citing the reference
file:line it follows (paths relative to /root/reference).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it, and only as the checker.  The product path
(``imagecompressionlearnedliftingandlearnedtreebasedmodels_amd``) never imports
from here and fails loudly when the HIP library is missing.

Parity pinning:
  * lifting / P-block / split+merge / MaskedConv2d / LowerBound /
    NonNegativeParametrizer / GDN / context-model wiring / RD loss:
    pinned against the reference's own Python run in the build container
    (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``).
  * compressai==1.2.1 leaf ops (GaussianConditional, EntropyBottleneck,
    RGB2YCbCr/YCbCr2RGB) and pytorch_wavelets (CDF 9/7, periodization):
    the packages are absent from the image and the reference holds no test
    vectors for them -> restated from their published algorithm,
    **parity unpinned** (checked against float64 closed forms and, for the
    CDF 9/7 filter bank, against the reference's own ``get_cdf97_filters``
    table and PyWavelets 1.1.1 of the local conda interpreter).
"""
