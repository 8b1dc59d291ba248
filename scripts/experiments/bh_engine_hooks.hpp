// bh_engine_hooks.hpp -- what -DBHGPU_EXPERIMENTS adds to the engine translation unit (scripts/ A/B builds only; the product
// library is built without the flag and carries an empty hook object instead: csrc/bh_engine.hip).
//
// BH_WALK_TIMELINE=<path>: every wavefront of the fp32 walk stamps its start and end -- s_memrealtime (the 100 MHz wall clock)
// AND s_memtime (shader clock cycles) --, its hardware id and its loop-iteration count; bh_destroy writes the last walk's
// records to <path> as 6 x uint64 per wave.  scripts/walk_timeline.py turns them into the occupancy profile of the launch and
// into the in-kernel clock, delta s_memtime / delta s_memrealtime x 100 MHz (MI355X_MICROARCH.md, DVFS give-back item 6).
#pragma once

#include <cstdio>
#include <cstdlib>
#include <vector>

struct ExpHooks {
    static constexpr int kWords = 6;
    uint64_t *timeline = nullptr;
    int64_t waves_cap = 0;

    hipError_t create(int64_t capacity)
    {
        if (!std::getenv("BH_WALK_TIMELINE")) return hipSuccess;
        waves_cap = capacity / bh::kWave + 8;
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&timeline), (size_t)kWords * waves_cap * sizeof(uint64_t));
        if (e != hipSuccess) return e;
        return hipMemset(timeline, 0, (size_t)kWords * waves_cap * sizeof(uint64_t));
    }
    void walk_args(bh::WalkFastArgs &a) const { a.timeline = timeline; }
    void destroy(int64_t n)
    {
        if (!timeline) return;
        const size_t words = (size_t)kWords * (size_t)((n + bh::kWave - 1) / bh::kWave);
        std::vector<uint64_t> h(words);
        if (hipMemcpy(h.data(), timeline, words * sizeof(uint64_t), hipMemcpyDeviceToHost) == hipSuccess) {
            if (FILE *fp = std::fopen(std::getenv("BH_WALK_TIMELINE"), "wb")) { std::fwrite(h.data(), 8, words, fp); std::fclose(fp); }
        }
        (void)hipFree(timeline);
        timeline = nullptr;
    }
};
