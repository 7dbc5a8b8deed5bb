#!/usr/bin/env python3
"""Generate the committed golden vectors by RUNNING THE REFERENCE's own object
code (oracle/_ref = /root/reference/convert.cpp + common.cpp compiled as they
lie, driven by oracle/ref_shim.cpp).  Container only: needs /root/reference.

    python tests/golden/make_golden.py

Writes tests/golden/*.npz (inputs + expected .yuv samples), index.json and
known_md5.json.  The npz files hold data only: input planes and output bytes.

known_md5.json: md5 of full-size .yuv frames.  The C1/C2/C3/C4/64x32 values
are the ones SURVEY.md section 8c recorded from the complete reference pipeline
(including tiff.cpp's write_yuv, which cannot be compiled in this image); this
script re-derives each with oracle/_ref + our restated write_yuv clamp and
refuses to write the file if any of them disagrees.
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import binding as ob  # noqa: E402

SURVEY_MD5 = {
    "C1_1080p_709_10b_fir": (dict(width=1920, height=1080, dst_depth=10, dst_matrix=ob.MATRIX_BT709, resampler=1),
                             "bfcececbcafb79ef14b46bf520689921"),
    "C1_1080p_709_10b_box": (dict(width=1920, height=1080, dst_depth=10, dst_matrix=ob.MATRIX_BT709, resampler=0),
                             "866cee72366f4b82a0c9f78ac64f0fda"),
    "C2_4k_2020_12b_fir": (dict(width=3840, height=2160, dst_depth=12, dst_matrix=ob.MATRIX_BT2020NC, resampler=1),
                           "1160366140de02df533d402243b39a5c"),
    "C2_4k_2020_12b_box": (dict(width=3840, height=2160, dst_depth=12, dst_matrix=ob.MATRIX_BT2020NC, resampler=0),
                           "775842f3d7ffbfbac19cb571cb2d082a"),
    "C3_4k_ydzdx_16b_444": (dict(width=3840, height=2160, dst_depth=16, dst_matrix=ob.MATRIX_YDZDX, chroma=ob.CHROMA_444),
                            "5388311273b7aca878b421173ea242f6"),
    "C4_8k_f16_2020_10b_fir": (dict(width=7680, height=4320, sample=ob.SAMPLE_F16, dst_depth=10,
                                    dst_matrix=ob.MATRIX_BT2020NC, resampler=1), "b7bf2b97ca54ad373f8b2520c6a33462"),
    "C4_8k_f16_2020_10b_box": (dict(width=7680, height=4320, sample=ob.SAMPLE_F16, dst_depth=10,
                                    dst_matrix=ob.MATRIX_BT2020NC, resampler=0), "48c8bed2f86abaea309d15aa119d40b7"),
    "tiny_64x32_2020_10b_fir": (dict(width=64, height=32, dst_depth=10, dst_matrix=ob.MATRIX_BT2020NC, resampler=1),
                                "15de7c9fc194f0c5eb2975d1874c9679"),
}


def md5(a):
    return hashlib.md5(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    o = ob.Oracle()
    r = ob.Ref()
    # ---- known answers at full size ------------------------------------
    known = {"source": "SURVEY.md 8c (complete reference pipeline); re-derived here with oracle/_ref", "cases": {}}
    for name, (kw, want) in SURVEY_MD5.items():
        d = ob.make_desc(**kw)
        planes = o.synth_frame(kw["width"], kw["height"], 0, f16=kw.get("sample") == ob.SAMPLE_F16)
        got = r.convert_frame(d, planes)
        if md5(got) != want:
            raise SystemExit(f"{name}: oracle/_ref gives {md5(got)}, SURVEY recorded {want}")
        known["cases"][name] = {"desc": kw, "md5": want, "bytes": int(got.nbytes)}
        print("known", name, want)
    with open(os.path.join(HERE, "known_md5.json"), "w") as f:
        json.dump(known, f, indent=1)

    # ---- small vectors, bytes committed ---------------------------------
    rng = np.random.default_rng(20261004)
    cases = []

    def emit(name, kw, planes):
        d = ob.make_desc(**kw)
        yuv = r.convert_frame(d, planes)
        fn = name + ".npz"
        np.savez_compressed(os.path.join(HERE, fn), in0=planes[0], in1=planes[1], in2=planes[2], yuv=yuv)
        cases.append({"file": fn, "desc": kw, "md5": md5(yuv)})
        print("vector", fn, md5(yuv))

    def rand(w, h, lo=0.0, hi=1.0):
        ps = [rng.uniform(lo, hi, w * h).astype(np.float32) for _ in range(3)]
        for p in ps:
            p[0], p[1] = lo, hi
            p[2] = 0.0 if lo == 0.0 else lo
        return ps

    emit("synth_64x32_2020_10b_fir", dict(width=64, height=32, dst_depth=10, dst_matrix=ob.MATRIX_BT2020NC, resampler=1),
         o.synth_frame(64, 32))
    emit("synth_64x32_2020_12b_box", dict(width=64, height=32, dst_depth=12, dst_matrix=ob.MATRIX_BT2020NC, resampler=0),
         o.synth_frame(64, 32, 1))
    emit("rand_96x40_709_10b_fir_full", dict(width=96, height=40, dst_depth=10, dst_matrix=ob.MATRIX_BT709, resampler=1,
                                            full_range=1), rand(96, 40))
    emit("rand_80x24_ydzdx_16b_444", dict(width=80, height=24, dst_depth=16, dst_matrix=ob.MATRIX_YDZDX,
                                         chroma=ob.CHROMA_444), rand(80, 24))
    emit("rand_72x20_y100_12b_box", dict(width=72, height=20, dst_depth=12, dst_matrix=ob.MATRIX_Y100, resampler=0),
         rand(72, 20))
    emit("rand_72x20_y500_14b_fir", dict(width=72, height=20, dst_depth=14, dst_matrix=ob.MATRIX_Y500, resampler=1),
         rand(72, 20))
    emit("rand_64x16_gbr_identity_10b_444", dict(width=64, height=16, dst_depth=10, dst_matrix=ob.MATRIX_GBR,
                                                 chroma=ob.CHROMA_444), rand(64, 16))
    emit("rand_64x16_stats_2_7", dict(width=64, height=16, dst_depth=12, dst_matrix=ob.MATRIX_BT2020NC, resampler=1),
         rand(64, 16, 2.0, 7.9))
    emit("f16_64x32_2020_10b_fir", dict(width=64, height=32, sample=ob.SAMPLE_F16, dst_depth=10,
                                       dst_matrix=ob.MATRIX_BT2020NC, resampler=1), o.synth_frame(64, 32, 2, f16=True))
    u16 = [rng.integers(0, 65536, 64 * 16).astype(np.uint16) for _ in range(3)]
    emit("u16_64x16_ydzdx_16to12b_fir", dict(width=64, height=16, sample=ob.SAMPLE_U16, src_depth=16, dst_depth=12,
                                            src_transfer=ob.TRANSFER_PQ, dst_transfer=ob.TRANSFER_PQ,
                                            dst_matrix=ob.MATRIX_YDZDX, resampler=1), u16)
    with open(os.path.join(HERE, "index.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_golden.py (oracle/_ref = reference object code)", "cases": cases}, f,
                  indent=1)


if __name__ == "__main__":
    main()
