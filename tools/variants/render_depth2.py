"""A/B of k_local_psf_render_wave: the weights of TWO pixels ahead in flight (three register sets,
the pixel loop fully unrolled) instead of one."""
import sys
from _edit import sub
root = sys.argv[1]
sub(root, "sdirt_render.hip", """    float wa[NI], ra[NI], wb[NI], rb[NI];
    load_w(x0 + wave, wa, ra);                       // in flight while the patch is staged""",
    """    float wl3[3][NI], wr3[3][NI];
    load_w(x0 + wave, wl3[0], wr3[0]);               // in flight while the patch is staged
    load_w(x0 + wave + NW, wl3[1], wr3[1]);""")
sub(root, "sdirt_render.hip", """#pragma unroll 1
    for (int j = 0; j < PPW; j += 2) {
        const int x = x0 + wave + j * NW;
        load_w(x + NW, wb, rb);
        pixel(x, wa, ra);
        if (j + 2 < PPW) load_w(x + 2 * NW, wa, ra);
        pixel(x + NW, wb, rb);
    }
}

// PSFNet.pred""", """#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int x = x0 + wave + j * NW;
        if (j + 2 < PPW) load_w(x + 2 * NW, wl3[(j + 2) % 3], wr3[(j + 2) % 3]);
        pixel(x, wl3[j % 3], wr3[j % 3]);
    }
}

// PSFNet.pred""")
