"""MI355X-native WaveNet-autoencoder training hot path.

Package layout:
  csrc/                 hand-written HIP kernels for gfx950 + the C-ABI (include/aewavenet.h)
  _lib.py               ctypes binding of the C-ABI shared library (fails loudly if missing)
  geometry.py           integer receptive-field algebra (restates reference vconv.py)
  config.py             hyper-parameter object + par/*.json key map
  engine.py             launch plans for encoder / bottleneck / decoder forward+backward
  autoencoder_model.py  drop-in ``AutoEncoder(hps)`` module surface
  mfcc_inverter.py      drop-in ``MfccInverter(hps)`` module surface
  optim.py              fused Adam on the flat parameter buffer
  dp.py                 data-parallel helpers (RCCL via torch.distributed)
"""
__version__ = "0.1.0"
