/* Sanitizer harness for the oracle (oracle/smallpt_oracle.c compiled with -fsanitize=address,undefined and OpenMP off):
 * a small Cornell-like render that exercises every material, the glass split stack, the depth cap and both cameras. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../oracle/smallpt_oracle.h"

static void sphere(orc_sphere* s, float r, float cx, float cy, float cz, float e, float c, int refl)
{
    memset(s, 0, sizeof *s);
    s->radius = r; s->center[0] = cx; s->center[1] = cy; s->center[2] = cz;
    s->emission[0] = s->emission[1] = s->emission[2] = e;
    s->color[0] = s->color[1] = s->color[2] = c;
    s->refl = refl;
}

int main(void)
{
    orc_sphere sc[6];
    sphere(&sc[0], 1e5f, 50, 1e5f, 81.6f, 0, .75f, ORC_DIFF);
    sphere(&sc[1], 1e5f, 50, -1e5f + 81.6f, 81.6f, 0, .75f, ORC_DIFF);
    sphere(&sc[2], 16.5f, 27, 16.5f, 47, 0, .999f, ORC_SPEC);
    sphere(&sc[3], 16.5f, 73, 16.5f, 78, 0, .999f, ORC_REFR);
    sphere(&sc[4], 600, 50, 681.6f - .27f, 81.6f, 1, 0, ORC_DIFF);
    sphere(&sc[5], 1000.0f, 50, 52, 200, 0, 1, ORC_SPEC);      /* p = 1 mirror shell: depth cap */
    orc_camera cam;
    orc_stats st;
    const uint32_t w = 24, h = 18;
    float* img = (float*)malloc(sizeof(float) * w * h * 3);
    orc_camera_smallpt(w, h, &cam);
    if (orc_render(sc, 5, &cam, w, h, 0, h, 40, 3, ORC_FLAG_NORMALISE, 1, img, &st)) return 1;   /* 40 samples: two D9 blocks */
    if (orc_render(sc, 6, &cam, 4, 4, 1, 2, 1, 0, 0, 1, img, &st)) return 1;
    if (st.max_depth_kills == 0) return 2;
    const float vx[3] = {1, 0, 0}, vy[3] = {0, 1, 0}, vz[3] = {0, 0, -1}, org[3] = {50, 52, 295.6f};
    orc_camera_pinhole(vx, vy, vz, org, 1.0f, &cam);
    if (orc_render(sc, 5, &cam, w, h, 0, h, 1, 7, 0, 1, img, &st)) return 1;
    if (orc_render(NULL, 0, &cam, 2, 2, 0, 2, 1, 0, 0, 1, img, &st)) return 1;
    printf("oracle sanitizer run ok: %llu bounces\n", (unsigned long long)st.bounces);
    free(img);
    return 0;
}
