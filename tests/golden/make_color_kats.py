"""Writes kat.json's "color" list: known answers of the colour neighbours (oracle/color_oracle.c).  COL-1, COL-3, COL-5 and
COL-7 are the hand-computed values tests/test_oracle.py has carried since round 1; COL-2, COL-4, COL-6 and COL-8 are the pixels a search
found to separate the mutants of tests/color_mutants.py that the old ones let through (the truncated fixed-point coefficients, green from
two roundings) and, for the 4:2:0 pairs, one small frame that separates all five mutants of each direction.  Answers: tests/golden/
derive_kats.py (plain integers, one pixel at a time).  PARITY UNPINNED like everything else: they pin the oracle to the constants
its header states, not to OpenCV.      python tests/golden/make_color_kats.py"""
import json
from pathlib import Path

import derive_kats as D

CASES = [
    dict(id="COL-1", op="bgr2yuv", src=[[255, 255, 255], [255, 0, 0], [0, 0, 0], [128, 128, 128], [0, 0, 255]],
         why="white, pure blue, black, grey, pure red (B,G,R): pure blue Y = (255*1868 + 8192) >> 14 = 29, U = ((226*8061) + (128<<14) + 8192) >> 14 = 239; pure red: V saturates at 255"),
    dict(id="COL-2", op="bgr2yuv", src=[[1, 1, 200], [235, 37, 0], [16, 255, 37], [0, 83, 254]],
         why="one pixel per truncated coefficient: (1,1,200) separates R2Y 4899 from 4898, (235,37,0) B2Y 1868 from 1867, (16,255,37) R2VI 14369 from 14368, (0,83,254) B2UI 8061 from 8060"),
    dict(id="COL-3", op="yuv2bgr", src=[[255, 128, 128], [29, 239, 103], [0, 128, 128], [128, 128, 128], [76, 91, 255]],
         why="COL-1's outputs taken back: white, blue, black and grey return exactly; pure red does not -- its V had saturated at 255 -- and returns as (1, 17, 221)"),
    dict(id="COL-4", op="yuv2bgr", src=[[37, 16, 100], [0, 0, 104], [0, 0, 153]],
         why="(37,16,100): green from ONE descale of the summed products, not two; (0,0,104) separates U2GI -6472 from -6471; (0,0,153) V2RI 18678 from 18677"),
    dict(id="COL-5", op="nv12_to_bgr", shape=[4, 2], src=[16, 235, 81, 0, 126, 81, 255, 16, 128, 128, 90, 240],
         why="video black / white; (110*1220542 + 2^19) >> 20 = 128; Y < 16 clamps to 0; the chroma of pure red (U=90, V=240) on the right block"),
    dict(id="COL-6", op="nv12_to_bgr", shape=[4, 2], src=[235, 0, 1, 16, 1, 235, 235, 100, 0, 0, 37, 64],
         why="one 4x2 frame that separates: full-range luma, missing luma clamp, missing rounding, NV21 order, chroma of the neighbouring pair"),
    dict(id="COL-7", op="bgr_to_nv12", shape=[2, 2], src=[0, 0, 255, 255, 255, 255, 0, 0, 0, 128, 128, 128],
         why="red, white / black, grey: (269484*255 + 2^19 + (16<<20)) >> 20 = 82; chroma from the top-left pixel (red): U 90, V 240"),
    dict(id="COL-8", op="bgr_to_nv12", shape=[2, 2], src=[128, 64, 16, 1, 128, 200, 0, 1, 64, 37, 235, 100],
         why="one 2x2 block that separates: averaged chroma, chroma of the bottom-right pixel, missing rounding, missing +16, V before U"),
]
out = []
for c in CASES:
    k = dict(c)
    k["dst"] = D.color_answer(c)
    out.append(k)
p = Path(__file__).parent / "kat.json"
kat = json.loads(p.read_text())
kat["_comment_color"] = ("Colour neighbours (SURVEY 8f N3, oracle/color_oracle.c): hand-computed values since round 1 (COL-1/3/5/7) plus one entry per family "
                         "added in round 5 so that every mutant of tests/color_mutants.py fails one (tests/test_color_kill_matrix.py).  Answers from "
                         "tests/golden/derive_kats.py.  Parity unpinned: pinned to the constants the oracle's header states, not to OpenCV.")
kat["color"] = out
p.write_text(json.dumps(kat, indent=1) + "\n")
for k in out:
    print(k["id"], k["op"], k["dst"])
