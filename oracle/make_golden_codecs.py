"""Golden vectors for the zarr chunk codecs xgcm_amd.io decodes itself (blosc container; numcodecs' zstd / lz4 framings).

Run in the build container: `python oracle/make_golden_codecs.py` -> tests/golden/codec_chunks.npz.  The compressed streams come
from the REAL libraries -- c-blosc 1.21 (the image's Anaconda tree, /opt/conda/lib/libblosc.so.1: the library numcodecs.Blosc
wraps, i.e. what `xarray.Dataset.to_zarr` writes by default), libzstd and liblz4 (system) -- called through ctypes; nothing of
xgcm_amd takes part.  Each case: the raw chunk bytes, the compressed bytes, and the `.zarray` "compressor" dict zarr would record."""
import ctypes
import ctypes.util
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "codec_chunks.npz")


def main():
    blosc = ctypes.CDLL(os.environ.get("XG_BLOSC_LIB", "/opt/conda/lib/libblosc.so.1"))
    blosc.blosc_compress_ctx.restype = ctypes.c_int
    blosc.blosc_compress_ctx.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p,
                                         ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int]
    blosc.blosc_get_version_string.restype = ctypes.c_char_p
    zstd = ctypes.CDLL(ctypes.util.find_library("zstd") or "libzstd.so.1")
    zstd.ZSTD_compress.restype = ctypes.c_size_t
    zstd.ZSTD_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    zstd.ZSTD_compressBound.restype = ctypes.c_size_t
    zstd.ZSTD_compressBound.argtypes = [ctypes.c_size_t]
    lz4 = ctypes.CDLL(ctypes.util.find_library("lz4") or "liblz4.so.1")
    lz4.LZ4_compress_default.restype = ctypes.c_int
    lz4.LZ4_compress_default.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    lz4.LZ4_compressBound.restype = ctypes.c_int
    lz4.LZ4_compressBound.argtypes = [ctypes.c_int]

    rng = np.random.default_rng(20261001)
    smooth = np.cumsum(rng.standard_normal((4, 30, 60)) * 1e-3, axis=2) + 15.0      # a model field: compresses, shuffle helps
    fields = {
        "f8_smooth": smooth,                                                        # 57.6 kB
        "f4_smooth": smooth.astype("<f4"),
        "f8_noise": rng.standard_normal((5, 7, 11)),                                # incompressible: blosc stores it (memcpy flag)
        "i2_ramp": (np.arange(3 * 50 * 33) % 977).astype("<i2").reshape(3, 50, 33),
        "u1_mask": (rng.random((4, 30, 31)) > 0.7).astype("u1"),                    # typesize 1: never shuffled
        "f8_tiny": np.array([[1.5, -2.25, np.nan]]),                                # below blosc's minimum: one leftover block
        "i8_big": (np.arange(30000, dtype="<i8") * 3) % 1013,                       # 240 kB: several blocks + a leftover one
    }
    out, cases = {}, []

    def blosc_case(name, a, cname, clevel, shuffle, blocksize=0):
        raw = np.ascontiguousarray(a).tobytes()
        dst = ctypes.create_string_buffer(len(raw) + 16 + 4 * 4096)
        n = blosc.blosc_compress_ctx(clevel, shuffle, a.dtype.itemsize, len(raw), raw, dst, len(dst), cname.encode(), blocksize, 1)
        assert n > 0, (name, cname, n)
        key = f"{name}__blosc_{cname}_{clevel}_{shuffle}_{blocksize}"
        out[key] = np.frombuffer(dst.raw[:n], dtype="u1")
        cases.append({"key": key, "field": name, "compressor": {"id": "blosc", "cname": cname, "clevel": clevel, "shuffle": shuffle, "blocksize": blocksize}})

    for name, a in fields.items():
        out["raw__" + name] = a
        blosc_case(name, a, "lz4", 5, 1)          # zarr's default: Blosc(cname="lz4", clevel=5, shuffle=SHUFFLE)
        blosc_case(name, a, "zstd", 3, 1)         # the usual choice of archived stores
        blosc_case(name, a, "zstd", 1, 2)         # bit shuffle
        blosc_case(name, a, "lz4hc", 4, 0)        # no shuffle
        blosc_case(name, a, "zlib", 2, 1)
        blosc_case(name, a, "lz4", 9, 1, 4096)    # forced small blocks: many splits
        blosc_case(name, a, "blosclz", 5, 1)      # blosc's own codec: refused by name unless libblosc itself is loadable
        raw = np.ascontiguousarray(a).tobytes()
        for level in (1, 7):                      # numcodecs.Zstd: one plain zstd frame
            dst = ctypes.create_string_buffer(zstd.ZSTD_compressBound(len(raw)))
            n = zstd.ZSTD_compress(dst, len(dst), raw, len(raw), level)
            key = f"{name}__zstd_{level}"
            out[key] = np.frombuffer(dst.raw[:n], dtype="u1")
            cases.append({"key": key, "field": name, "compressor": {"id": "zstd", "level": level}})
        dst = ctypes.create_string_buffer(lz4.LZ4_compressBound(len(raw)))   # numcodecs.LZ4: uint32 LE raw size + one lz4 block
        n = lz4.LZ4_compress_default(raw, dst, len(raw), len(dst))
        key = f"{name}__lz4"
        out[key] = np.frombuffer(len(raw).to_bytes(4, "little") + dst.raw[:n], dtype="u1")
        cases.append({"key": key, "field": name, "compressor": {"id": "lz4", "acceleration": 1}})
    out["cases"] = np.frombuffer(json.dumps({"blosc_version": blosc.blosc_get_version_string().decode(), "cases": cases}).encode(), dtype="u1")
    np.savez_compressed(OUT, **out)
    print(f"{OUT}: {len(cases)} cases, {os.path.getsize(OUT)} bytes, c-blosc {blosc.blosc_get_version_string().decode()}")


if __name__ == "__main__":
    sys.exit(main())
