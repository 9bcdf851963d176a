// deal_main.cpp -- CPU check of spt_deal.h: for a set of task counts, the queue positions 0 .. 64 * ceil(ntasks / 64) - 1 map onto every
// task id 0 .. ntasks - 1 exactly once, every other position (the holes of the last stride, anything at or beyond the end, the
// grid-pool kernel's "nothing left" position 0xFFFFFF00) maps to "no task", and the 64 tasks of a chunk lie ceil(ntasks / 64) apart.
#include <cstdio>
#include <utility>
#include <vector>
#include "../../optix-test-smallpt_amd/csrc/spt_deal.h"

int main()
{
    const uint32_t counts[] = {1, 2, 63, 64, 65, 127, 128, 129, 7560, 7844, 4096, 4097, 100000, 393216 * 64 + 17};
    for (uint32_t n : counts) {
        const uint32_t nch = (n + 63u) >> 6;
        std::vector<unsigned char> seen(n, 0);
        unsigned long long valid = 0;
        for (uint64_t q = 0; q < (uint64_t)nch * 64u + 256u; ++q) {
            const uint32_t t = deal_task((uint32_t)q, n);
            if (t == 0xFFFFFFFFu) continue;
            if (t >= n || q >= (uint64_t)nch * 64u) { std::printf("ntasks %u: position %llu maps to %u\n", n, (unsigned long long)q, t); return 1; }
            if (seen[t]++) { std::printf("ntasks %u: task %u handed out twice (position %llu)\n", n, t, (unsigned long long)q); return 1; }
            if ((q & 63u) != 0u && t - deal_task((uint32_t)q - 1u, n) != nch && deal_task((uint32_t)q - 1u, n) != 0xFFFFFFFFu) { std::printf("ntasks %u: stride\n", n); return 1; }
            ++valid;
        }
        if (valid != n) { std::printf("ntasks %u: %llu tasks handed out\n", n, valid); return 1; }
        for (uint32_t q : {0xFFFFFF00u, 0xFFFFFF3Fu, 0xFFFFFFFFu, n >= 64u ? nch * 64u : 64u})
            if (deal_task(q, n) != 0xFFFFFFFFu) { std::printf("ntasks %u: position %u beyond the end maps to a task\n", n, q); return 1; }
    }
    // deal_task_tiles: every task exactly once below deal_tiles_end(), nothing at or beyond it, a chunk's tasks are the pixels of one
    // 8 x 8 tile with one sub-index, a pixel's S tasks lie G chunks apart
    for (uint32_t S : {4u, 8u, 32u})
        for (auto wh : {std::pair<uint32_t, uint32_t>{1, 1}, {7, 9}, {8, 8}, {9, 17}, {64, 1}, {40, 30}, {1280, 720}, {1023, 3}}) {
            const uint32_t w = wh.first, rows = wh.second, n = w * rows * S, end = deal_tiles_end(w, rows, S), gx = (w + 7u) >> 3, G = gx * ((rows + 7u) >> 3);
            std::vector<unsigned char> seen(n, 0);
            unsigned long long valid = 0;
            for (uint64_t q = 0; q < (uint64_t)end + 256u; ++q) {
                const uint32_t t = deal_task_tiles((uint32_t)q, w, rows, S);
                if (t == 0xFFFFFFFFu) continue;
                if (t >= n || q >= end) { std::printf("tiles %u x %u x %u: position %llu maps to %u\n", w, rows, S, (unsigned long long)q, t); return 1; }
                if (seen[t]++) { std::printf("tiles %u x %u x %u: task %u handed out twice\n", w, rows, S, t); return 1; }
                const uint32_t c = (uint32_t)q >> 6, l = (uint32_t)q & 63u, pix = t / S, x = pix % w, y = pix / w, g = c % G;
                if (t % S != c / G || x != (g % gx) * 8u + (l & 7u) || y != (g / gx) * 8u + (l >> 3)) { std::printf("tiles %u x %u x %u: position %llu is task %u\n", w, rows, S, (unsigned long long)q, t); return 1; }
                ++valid;
            }
            if (valid != n) { std::printf("tiles %u x %u x %u: %llu tasks handed out\n", w, rows, S, valid); return 1; }
        }
    std::printf("deal_task ok\n");
    return 0;
}
