"""Host pacing of process_device: time of each call (no sync) against the GPU's time per block.
python profiles/tools/hostpace.py [receivers] [rate] [mix] [blocks]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from _hip import DevBuf, sync
from quisk_amd import Engine, rx_filter
from quisk_amd.synth import tune_offsets, mode_mix
nrx = int(sys.argv[1]) if len(sys.argv) > 1 else 128
fs = int(sys.argv[2]) if len(sys.argv) > 2 else 61440000
mix = sys.argv[3] if len(sys.argv) > 3 else "mixed"
K = int(sys.argv[4]) if len(sys.argv) > 4 else 24
block = 1 << 20
rng = np.random.default_rng(5)
x = np.round((rng.standard_normal(2 * block) + 1j * rng.standard_normal(2 * block)) * 2.0 ** 20)
dev = DevBuf(x)
modes = mode_mix(nrx, mix)
tunes = tune_offsets(nrx, fs)
fftN = int(os.environ.get("QK_FFT", "0"))            # QK_FFT=1048576: a windowed FFT frame of every block beside the receivers
kw = dict(fft_size=fftN, data_width=4096) if fftN else {}
eng = Engine(fs, n_receivers=nrx, max_block=block, fp32=bool(int(os.environ.get("QK_FP32", "0"))), **kw)
for r in range(nrx):
    fi, fq, bw, off = rx_filter(fs, int(modes[r]))
    eng.set_mode(r, int(modes[r])); eng.set_tune(r, int(tunes[r])); eng.set_filters(r, fi, fq, bw, off)
for i in range(4):
    eng.process_device(dev.at((i % 2) * block * 16), block)
eng.sync()
ts = []
t0 = time.perf_counter()
for i in range(K):
    a = time.perf_counter()
    eng.process_device(dev.at((i % 2) * block * 16), block)
    ts.append(time.perf_counter() - a)
t_enq = time.perf_counter() - t0
eng.sync()
t_all = time.perf_counter() - t0
ts = np.array(ts) * 1e3
print("host per call ms: median %.3f min %.3f max %.3f; enqueue total %.2f ms, with GPU %.2f ms (%.3f ms/block)"
      % (np.median(ts), ts.min(), ts.max(), t_enq * 1e3, t_all * 1e3, t_all / K * 1e3))
print(" ".join("%.2f" % t for t in ts))
eng.close()
