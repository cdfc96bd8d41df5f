/* A plain C99 client of the HOST-ONLY entries of the boundary (no GPU needed): the layout rule of the multi-GPU gather
 * (qil_sweep_unshuffle), qil_host_cpu_budget, qil_version and the error convention.  Compiled, linked against
 * libqilhip.so and RUN by the CPU suite (tests/test_cabi_symbols.py). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "qilaplace_hip.h"

int main(void) {
    /* 3 ranks, 7 items of width 2: rank r contributes items r, r + 3, ... in ceil(7 / 3) = 3 slots */
    enum { WORLD = 3, ITEMS = 7, WIDTH = 2, PER = 3 };
    double gathered[WORLD * PER * WIDTH * 2];
    double out[ITEMS * WIDTH * 2];
    int r, slot, w, i, budget = 0;
    memset(gathered, 0, sizeof gathered);
    for (r = 0; r < WORLD; ++r)
        for (slot = 0, i = r; i < ITEMS; ++slot, i += WORLD)
            for (w = 0; w < WIDTH; ++w) {
                gathered[2 * (WIDTH * (PER * r + slot) + w)] = 100.0 * i + w;        /* re */
                gathered[2 * (WIDTH * (PER * r + slot) + w) + 1] = -(100.0 * i + w); /* im */
            }
    if (qil_sweep_unshuffle(WORLD, ITEMS, WIDTH, gathered, out) != QIL_OK) {
        fprintf(stderr, "unshuffle failed: %s\n", qil_last_error());
        return 1;
    }
    for (i = 0; i < ITEMS; ++i)
        for (w = 0; w < WIDTH; ++w)
            if (out[2 * (WIDTH * i + w)] != 100.0 * i + w || out[2 * (WIDTH * i + w) + 1] != -(100.0 * i + w)) {
                fprintf(stderr, "item %d entry %d misplaced\n", i, w);
                return 2;
            }
    /* error convention: a bad argument returns a status and leaves a message */
    if (qil_sweep_unshuffle(0, ITEMS, WIDTH, gathered, out) == QIL_OK || strlen(qil_last_error()) == 0) return 3;
    if (qil_host_cpu_budget(&budget) != QIL_OK || budget < 1) return 4;
    printf("%s budget %d OK\n", qil_version(), budget);
    return 0;
}
