// spt_deal.h -- which task a queue position stands for (the persistent kernels other than the pool kernel); shared with the CPU check
// of the mapping (tests/sanitize/deal_main.cpp).
#ifndef SPT_DEAL_H
#define SPT_DEAL_H
#include <stdint.h>
#if defined(__HIPCC__)
#define SPT_DEAL_HD __host__ __device__ __forceinline__
#else
#define SPT_DEAL_HD inline
#endif

// ---- task dealing (the persistent kernels other than the pool kernel) ----
// The task queue hands out positions q = 0, 1, 2, ... (in chunks of 64 per wave); position q stands for task
// (q >> 6) + (q & 63) * ceil(ntasks / 64): the 64 tasks of a chunk lie a 64th of the launch apart instead of side by side.  Task ids
// are ((pixel * 4 + cell) << nb_log2) | block, so consecutive ids are the sample blocks of ONE pixel -- and the cost of a block is a
// property of what its pixel looks at: a pixel that sees a closed mirror ball has every sample run to the depth cap.  Dealt out side by
// side such a pixel's blocks land in one wave, which then drags 16 deep paths through its phases long after the others have finished
// (profiles/r04_fuzz_deep_cases.txt); dealt out with the stride they go to 16 waves.  A bijection of the valid positions onto
// 0 .. ntasks - 1 (mixed radix), 0xFFFFFFFF = no task.  Which wave runs a task never changes its result.
SPT_DEAL_HD uint32_t deal_task(uint32_t q, uint32_t ntasks)
{
    const uint32_t nch = (ntasks + 63u) >> 6;
    const uint32_t t = (q >> 6) + (q & 63u) * nch;
    return ((q >> 6) < nch && t < ntasks) ? t : 0xFFFFFFFFu;
}

// ---- the triangle hierarchy's kernel deals the other way round in one respect (scenes without mirror / glass materials): the 64 tasks
// of a chunk belong to an 8 x 8 TILE of pixels -- their rays visit the same nodes, the per-lane walks of a wave have similar lengths and
// the node loads hit the cache (the shipped mesh scene at 1280 x 720: 4.4 -> 1.5 ms per frame; 64 x 1 strips: 1.7) -- while the S = 4 * NB
// tasks of ONE pixel still lie far apart in the queue: chunk c stands for (sub, tile) = (c / G, c % G), G = ceil(w / 8) * ceil(rows / 8)
// (w x rows pixels, row-major, pixel ids as in the task id), and its lane l for the tile's pixel (l & 7, l >> 3), task (y w + x) S + sub.
// Positions of a tile's part beyond the image are holes (0xFFFFFFFF) INSIDE the range -- the caller fetches again --; everything at or
// beyond deal_tiles_end() is the end.
SPT_DEAL_HD uint32_t deal_tiles_end(uint32_t w, uint32_t rows, uint32_t S) { return ((w + 7u) >> 3) * ((rows + 7u) >> 3) * 64u * S; }
SPT_DEAL_HD uint32_t deal_task_tiles(uint32_t q, uint32_t w, uint32_t rows, uint32_t S)
{
    const uint32_t gx = (w + 7u) >> 3, G = gx * ((rows + 7u) >> 3);
    const uint32_t c = q >> 6, l = q & 63u;
    const uint32_t sub = c / G, g = c - sub * G;
    const uint32_t ty = g / gx, tx = g - ty * gx;
    const uint32_t x = (tx << 3) + (l & 7u), y = (ty << 3) + (l >> 3);
    return (sub < S && x < w && y < rows) ? (y * w + x) * S + sub : 0xFFFFFFFFu;
}

#endif
