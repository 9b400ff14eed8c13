"""Writes the "derived" section of kat.json: one known answer per quirk of SURVEY.md Appendix A that the kill matrix
(tests/test_kat_kill_matrix.py) found unguarded in round 5.  Inputs: CL-6..CL-10 and EQ-7 were designed by hand (their `why` is the
hand derivation); the others are the smallest inputs a random search found on which ONE mutant of tests/oracle_mutants.py and the
restatement disagree.  ANSWERS: every `dst` below comes from tests/golden/derive_kats.py -- Appendix A in exact rational arithmetic
with explicit binary32 roundings, no code shared with oracle/ -- and `why` quotes its trace of the deciding pixel.  Neither OpenCV
(absent) nor the oracle produced them; the test then checks that BOTH oracles and the HIP kernels reproduce them.

    python tests/golden/make_derived_kats.py        (rewrites kat.json's "derived" list; idempotent)
"""
import json
from pathlib import Path

import derive_kats as D

HERE = Path(__file__).parent
CASES = [
    dict(id="EQ-7", op="equalize", shape=[1, 15], src=[0] + [1] * 7 + [2] * 7, guards=["eq_scale_double"],
         hand="total - hist[first] = 14; scale = rn32(255/14) = 18.214285 (binary32, BELOW 255/14); lut[1] = cvRound(rn32(7 * scale)): in reals "
              "7 * 255/14 = 127.5 exactly (a tie -> 128), but the binary32 scale is low, the product rounds to just under 127.5 -> 127. "
              "A double (or exact) evaluation gives 128."),
    dict(id="CL-6", op="clahe", shape=[1, 3], src=[10, 30, 20], clip=0.0, tiles=[2, 1], deciding_pixel=[0, 2], guards=["cl_pad_reflect"],
         hand="W=3 not divisible by 2 -> right pad 1 AND bottom pad 1 (H % 1 == 0, still padded): ext 4x2, tile 2x2, area 4, lutScale 63.75. "
              "REFLECT_101: column 3 reads column 1 (30); the single row repeats.  tile0 = {10,30}x2: LUT[10]=cvRound(2*63.75=127.5)=128 (even), "
              "LUT[20]=128, LUT[30]=255.  tile1 = {20,30}x2: LUT[20]=128, LUT[30]=255.  x=2 (v=20): txf=0.5, xa=0.5 -> 0.5*128+0.5*128 = 128.  "
              "BORDER_REFLECT would pad with 20: tile1 = {20,20}x2, LUT[20]=255 -> 0.5*128+0.5*255 = 191.5 -> 192."),
    dict(id="CL-7", op="clahe", shape=[16, 16], src=[0] * 256, clip=2.9999999, tiles=[1, 1], guards=["cl_clip_float"],
         hand="area 256: clip = (int)(2.9999999 * 256 / 256) in DOUBLE = 2 (2.9999999f would be 3.0f -> 3).  h[0]=256 -> 2, clipped 254, batch 0, "
              "resid 254, step 1: bins 0..253 += 1 -> h[0]=3; LUT[0] = cvRound(3 * 255/256 = 2.988) = 3.  With clip 3: h[0]=4 -> 4."),
    dict(id="CL-8", op="clahe", shape=[12, 16], src=[0] * 192, clip=2.0, tiles=[1, 1], guards=["cl_clip_round"],
         hand="area 192: clip = (int)(2.0 * 192 / 256 = 1.5) = 1 (truncation; rounding gives 2).  h[0]=192 -> 1, clipped 191, resid 191, step 1: "
              "bins 0..190 += 1 -> h[0]=2; LUT[0] = cvRound(2 * 255/192 = 2.656) = 3.  With clip 2: h[0]=3 -> cvRound(3.98) = 4."),
    dict(id="CL-9", op="clahe", shape=[16, 16], src=[2] * 6 + [5] * 250, clip=2.0, tiles=[1, 1], guards=["cl_redistribute_until_stable"],
         hand="area 256, clip 2: h[2]=6->2, h[5]=250->2, clipped 252, batch 0, resid 252, step 1: bins 0..251 += 1 -> h[2]=h[5]=3 (ABOVE the clip, "
              "and they stay there: ONE pass).  cum(2)=1+1+3=5 -> cvRound(5*255/256=4.98)=5; cum(5)=5+1+1+3=10 -> cvRound(9.96)=10.  "
              "Redistributing again would take 1 from bins 2 and 5 and give it to bins 0 and 128: cum(5)=9 -> 9."),
    dict(id="CL-10", op="clahe", shape=[16, 16], src=[100] * 256, clip=156.0, tiles=[1, 1], guards=["cl_residual_step_ceil", "cl_residual_first_bins"],
         hand="area 256, clip 156: h[100]=256->156, clipped 100, batch 0, resid 100, step = 256/100 = 2 (truncated): bins 0,2,...,198 += 1.  "
              "cum(100) = 51 (bins 0,2,..,100) + 156 = 207 -> cvRound(207*255/256 = 206.19) = 206.  step 3 would reach 34 bins <= 100 -> 189; "
              "'the first 100 bins' -> 255."),
]
FOUND = {   # smallest disagreements a random search found (gpurun_out scratch script, round 5); answers derived below, not taken from the search
    "cl_lut_scale_double": dict(shape=[1, 13], src=[9, 9, 91, 9, 91, 9, 91, 91, 9, 91, 9, 91, 9], clip=2.0, tiles=[2, 1], deciding_pixel=[0, 2]),
    "cl_weights_after_clamp": dict(shape=[2, 3], src=[48, 208, 208, 208, 143, 48], clip=0.0, tiles=[2, 1], deciding_pixel=[1, 2]),
    "cl_coord_fma": dict(shape=[13, 1], src=[89, 89, 89, 89, 89, 81, 89, 81, 81, 81, 89, 89, 89], clip=0.0, tiles=[1, 2], deciding_pixel=[7, 0]),
    "cl_coord_divide": dict(shape=[1, 8], src=[152, 152, 2, 152, 152, 152, 2, 2], clip=0.0, tiles=[3, 1], deciding_pixel=[0, 7]),
    "cl_blend_fma": dict(shape=[7, 2], src=[174, 154, 174, 174, 174, 154, 154, 174, 154, 154, 154, 154, 174, 174], clip=0.0, tiles=[1, 3], deciding_pixel=[5, 0]),
    "cl_blend_y_first": dict(shape=[9, 2], src=[10, 10, 10, 5, 5, 5, 10, 5, 10, 5, 10, 5, 5, 10, 10, 10, 10, 10], clip=0.0, tiles=[2, 3], deciding_pixel=[2, 1]),
    "cl_blend_double": dict(shape=[2, 5], src=[4, 4, 4, 4, 4, 210, 4, 4, 210, 210], clip=0.0, tiles=[2, 1], deciding_pixel=[0, 2]),
    "cl_blend_round_half_up": dict(shape=[5, 1], src=[201, 93, 93, 93, 93], clip=0.0, tiles=[1, 2], deciding_pixel=[3, 0]),
    "cl_tile_size_from_unpadded": dict(shape=[1, 3], src=[228, 20, 20], clip=0.0, tiles=[2, 1], deciding_pixel=[0, 1]),
}
for n, (m, c) in enumerate(FOUND.items()):
    CASES.append(dict(id=f"CL-{11 + n}", op="clahe", guards=[m], **c))

out = []
for c in CASES:
    tr = []
    dst = D.derived_answer(c, tr)
    k = {key: c[key] for key in ("id", "op", "shape", "src", "clip", "tiles", "deciding_pixel", "guards") if key in c}
    k["dst"] = [v for row in dst for v in row]
    k["why"] = (c["hand"] + "  || exact-arithmetic trace: " if "hand" in c else "exact-arithmetic trace (derive_kats.py): ") + " | ".join(tr)
    out.append(k)
p = HERE / "kat.json"
kat = json.loads(p.read_text())
kat["_comment_derived"] = ("Round 5: one known answer per quirk of App. A that no earlier KAT discriminated (tests/test_kat_kill_matrix.py prints "
                           "the mutant x KAT matrix).  `dst` comes from tests/golden/derive_kats.py (exact rationals + explicit binary32 roundings), "
                           "NOT from OpenCV (absent here) and NOT from the oracle; `guards` names the mutant(s) of tests/oracle_mutants.py it was made for.")
kat["derived"] = out
p.write_text(json.dumps(kat, indent=1) + "\n")
for k in out:
    print(k["id"], k["guards"], k["dst"][:16], "...")
