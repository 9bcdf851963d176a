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

#endif
