// vb_prof.h -- opt-in per-STREAM launch timing behind vb_stream_profile (include/visualbert_hip.h): a stream that asked for
// it gets a HIP event pair around every GEMM launch enqueued on it; other streams are untouched.  The recorder of the stream
// being served is looked up once per extern "C" entry (vb_prof_select), like the launch options.  The kernel-logic simulator
// build (-DVB_EMU, host C++) has no events: its stubs are here so that no kernel file needs a conditional.
#pragma once
#include "vb_rt.h"
#include <stdint.h>
// a caller that runs ONE algorithmic product as several launches (the split-operand weight gradient: three plane pairs) scales
// the recorded FLOPs so that their sum is the algorithmic figure, and tags the records (key bit 256 = split operands)
static thread_local double t_vb_prof_scale = 1.0;
static thread_local int t_vb_prof_key_or = 0;
#ifndef VB_EMU
#include <mutex>
#include <utility>
#include <vector>

struct VbProfRec { hipEvent_t e0, e1; double flops; int key; };
static std::mutex g_vb_prof_mutex;
static std::vector<std::pair<void*, std::vector<VbProfRec>*>> g_vb_prof_table;
static thread_local std::vector<VbProfRec>* t_vb_prof = nullptr;      // recorder of the stream this thread is serving

static inline std::vector<VbProfRec>* vb_prof_for(void* stream) {
    std::lock_guard<std::mutex> lock(g_vb_prof_mutex);
    for (auto& e : g_vb_prof_table) if (e.first == stream) return e.second;
    return nullptr;
}
static inline void vb_prof_select(void* stream) { t_vb_prof = vb_prof_for(stream); }

// run `launch` (which enqueues exactly one kernel on `stream`), bracketed by events when the stream is being profiled
template <typename F>
static inline int vb_prof_launch(double flops, int key, hipStream_t stream, F&& launch) {
    if (t_vb_prof) {
        VbProfRec r;
        if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return VB_ERR_LAUNCH;
        r.flops = flops * t_vb_prof_scale; r.key = key | t_vb_prof_key_or;
        (void)hipEventRecord(r.e0, stream);
        launch();
        (void)hipEventRecord(r.e1, stream);
        t_vb_prof->push_back(r);
        return vb_check_launch();
    }
    launch();
    return vb_check_launch();
}
// forget the record vb_prof_launch just made (the launch turned out to be a no-op: the caller launches something else instead)
static inline void vb_prof_drop_last() {
    if (t_vb_prof && !t_vb_prof->empty()) {
        (void)hipEventDestroy(t_vb_prof->back().e0); (void)hipEventDestroy(t_vb_prof->back().e1);
        t_vb_prof->pop_back();
    }
}
static inline int vb_prof_enable(void* stream, int enable) {
    std::lock_guard<std::mutex> lock(g_vb_prof_mutex);
    for (size_t i = 0; i < g_vb_prof_table.size(); ++i)
        if (g_vb_prof_table[i].first == stream) {
            for (auto& r : *g_vb_prof_table[i].second) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
            delete g_vb_prof_table[i].second;
            g_vb_prof_table.erase(g_vb_prof_table.begin() + i);
            break;
        }
    if (enable) g_vb_prof_table.emplace_back(stream, new std::vector<VbProfRec>());
    return VB_OK;
}
static inline int64_t vb_prof_read(void* stream, double* ms, double* flops, int* key, int64_t max_records) {
    std::vector<VbProfRec>* recs = vb_prof_for(stream);
    if (!recs) return 0;
    int64_t n = 0;
    for (auto& r : *recs) {
        if (n >= max_records) break;
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.e0, r.e1) != hipSuccess) return -1;   // caller must synchronise first
        if (ms) ms[n] = t;
        if (flops) flops[n] = r.flops;
        if (key) key[n] = r.key;
        ++n;
    }
    return n;
}
#else
static inline void vb_prof_select(void*) {}
template <typename F> static inline int vb_prof_launch(double, int, hipStream_t, F&& launch) { launch(); return vb_check_launch(); }
static inline void vb_prof_drop_last() {}
static inline int vb_prof_enable(void*, int) { return VB_OK; }
static inline int64_t vb_prof_read(void*, double*, double*, int*, int64_t) { return 0; }
#endif
