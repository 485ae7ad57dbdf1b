"""config 1 (1 USB receiver, 192 ksps, 960-sample host blocks, audio pulled every block): wall time per block."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from quisk_amd import Engine, rx_filter
from quisk_amd.synth import make_iq
fs, blk, nblk = 192000, 960, int(sys.argv[1]) if len(sys.argv) > 1 else 300
x = make_iq(blk * nblk, fs, [19200], [3], 1001)
eng = Engine(fs, n_receivers=1, max_block=blk, fft_size=4096, data_width=4096)
fi, fq, bw, off = rx_filter(fs, 3)
eng.set_mode(0, 3); eng.set_tune(0, 19200); eng.set_filters(0, fi, fq, bw, off)
out = np.empty(eng.out_cap, dtype=np.complex128)
for i in range(20):
    eng.process(x[i * blk:(i + 1) * blk])
eng.sync()
tp = tq = 0.0
t0 = time.perf_counter()
for i in range(20, nblk):
    a = time.perf_counter()
    eng.process(x[i * blk:(i + 1) * blk])
    b = time.perf_counter()
    eng.L.qk_pull_audio(eng.h, 0, out.ctypes.data, eng.out_cap)
    c = time.perf_counter()
    tp += b - a; tq += c - b
dt = time.perf_counter() - t0
print("per block: %.1f us (process call %.1f us, pull_audio %.1f us)" % (dt / (nblk - 20) * 1e6, tp / (nblk - 20) * 1e6, tq / (nblk - 20) * 1e6))
eng.close()
