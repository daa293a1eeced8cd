"""A/B of k_local_psf_render_wave / k_psfnet_render_wave: patch rows padded by one position -- the
last tap of a kernel row and the first of the next then fall into different LDS banks (84 - 20 = 64 eight-byte words
apart today: the same bank pair; SQ_LDS_BANK_CONFLICT = 47 % of SQ_LDS_IDX_ACTIVE, profiles/r06/summary_render.json)."""
import sys
from _edit import sub
root = sys.argv[1]
sub(root, "sdirt_render.hip", "    constexpr int NPOS = KS * PW;                    // patch positions",
    "    constexpr int NPOS = KS * PW;                    // patch positions\n    constexpr int PWS = PW + 1;                      // row stride in LDS")
sub(root, "sdirt_render.hip", "    constexpr int NPOS = KS * PW;\n", "    constexpr int NPOS = KS * PW;\n    constexpr int PWS = PW + 1;\n")
sub(root, "sdirt_render.hip", "        ptap[it] = (KS - 1 - fi) * PW + (KS - 1 - fj);", "        ptap[it] = (KS - 1 - fi) * PWS + (KS - 1 - fj);", count=2)
sub(root, "sdirt_render.hip", "                patch[e] = v;", "                patch[r * PWS + (e - r * PW)] = v;", count=2)
sub(root, "sdirt_render.hip", "return (size_t)21 * (kChunk + 20) * 4 * (hf ? 2 : 4);", "return (size_t)21 * (kChunk + 21) * 4 * (hf ? 2 : 4);")
sub(root, "sdirt_render.hip", "(size_t)21 * 84 * 8, st>>>(img, rl, rr, H, W, out_l, out_r);", "(size_t)21 * 85 * 8, st>>>(img, rl, rr, H, W, out_l, out_r);")
