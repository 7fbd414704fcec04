// A second HIP stream per device (shared by the evaluation and the training GRU, towers.hip / gru_train.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace itr {

// A second HIP stream per device for the reverse direction of a bi-GRU: the two recurrences are independent, and a time step's
// GEMM (M = active captions <= a few thousand rows) leaves most CUs idle in its last wave of tiles -- the other direction fills them.
// The stream is created once per device (under a lock: several host threads may encode at once); the fork / join events are
// per call, so concurrent callers never share an event.
struct SideStream {
    hipStream_t st = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    bool forked = false;
    // Makes `main` wait for everything queued on the side stream.  Called on EVERY path out of itr_gru_fwd after the fork, the
    // error paths included: the caller frees the workspace after an error, and kernels still queued on the side stream would
    // read and write recycled memory.
    void join_into(hipStream_t main) {
        if (forked) {
            if (hipEventRecord(join, st) != hipSuccess || hipStreamWaitEvent(main, join, 0) != hipSuccess) (void)hipStreamSynchronize(st);
            forked = false;
        }
    }
    ~SideStream() {
        if (fork) (void)hipEventDestroy(fork);
        if (join) (void)hipEventDestroy(join);
    }
};
constexpr int GRU_MAX_CHAINS = 4;      // interleaved caption chains of the last-state recurrence (below): the caller's stream + 3
bool side_stream(SideStream &s, int k = 0);      // towers.hip

#ifdef ITR_SIDE_STREAM_IMPL
bool side_stream(SideStream &s, int k) {
    static std::mutex mu;
    static hipStream_t per_dev[16][GRU_MAX_CHAINS - 1] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16 || k < 0 || k >= GRU_MAX_CHAINS - 1) return false;
    {
        std::lock_guard<std::mutex> lock(mu);
        if (!per_dev[dev][k] && hipStreamCreateWithFlags(&per_dev[dev][k], hipStreamNonBlocking) != hipSuccess) { per_dev[dev][k] = nullptr; return false; }
        s.st = per_dev[dev][k];
    }
    return hipEventCreateWithFlags(&s.fork, hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&s.join, hipEventDisableTiming) == hipSuccess;
}
#endif

}  // namespace itr
