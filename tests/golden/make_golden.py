"""Generates tests/golden/oracle_vectors.npz from the CPU oracle (oracle/sdr_oracle.cpp).

The reference itself cannot be built in this image (needs <expected>, C++23), so these are
regression pins of the oracle, not reference outputs; the reference-held vectors (its gtest
inputs and expectations) are restated directly in tests/test_oracle_pins.py.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import binding as o  # noqa: E402


def main():
    rng = np.random.default_rng(20261111)
    out = {}
    for n in (8, 64, 1024):
        x = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
        out[f"fft_in_c64_{n}"] = x
        out[f"fft_out_c64_{n}"] = o.fft2(x)[1]
        out[f"ifft_out_c64_{n}"] = o.fft2(x, inverse=True)[1]
        xd = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex128)
        out[f"fft_in_c128_{n}"] = xd
        out[f"fft_out_c128_{n}"] = o.fft2(xd)[1]
    out["twiddles_c64_1024"] = o.twiddles(1024)
    out["twiddles_inv_c64_64"] = o.twiddles(64, inverse=True)
    # demapper: random points plus points on / next to decision boundaries and far outliers
    for bps in (2, 4, 6, 8):
        nrm, inrm, tab = o.constellation(bps)
        lv = np.unique(tab.real)
        edges = np.concatenate([(lv[:-1] + lv[1:]) / 2, lv, [lv[0] - 7, lv[-1] + 9, 5000.0, -3.0e4]]).astype(np.float32) * nrm
        gx, gy = np.meshgrid(edges, edges)
        pts = (gx + 1j * gy).astype(np.complex64).ravel()
        pts = np.concatenate([pts, np.nextafter(pts.real, np.float32(9)) + 1j * pts.imag,
                              ((rng.standard_normal(256) + 1j * rng.standard_normal(256)) * 0.9).astype(np.complex64)]).astype(np.complex64)
        out[f"demap_pts_{bps}"] = pts
        out[f"demap_idx_{bps}"] = o.nearest(bps, pts)
        out[f"demap_llr_{bps}"] = o.soft_demap(bps, pts, 1.75)
    rx, bits = o.link_generate(64, 16, 2, 1, 0, 3, 0.05)
    out["link_rx_64_qpsk"] = rx
    out["link_bits_64_qpsk"] = bits
    out["link_ber_1024_16qam"] = o.link_ber(1024, 256, 4, 2, 28, 2, 0.012, 9.0)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_vectors.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
