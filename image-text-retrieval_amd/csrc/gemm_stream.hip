// Streaming fp32 "NT" GEMM for SHORT K:  C[M,N] = act(A[M,K] * B[N,K]^T + bias[N]),  act in {none, relu, gelu}.
//
// gemm_nt_fast_kernel (gemm_f32.hip) works tile by tile: every 128 x 128 output tile pays a pipeline fill (its first
// operand chunk costs a full L2 / HBM latency before the first MFMA) and a drain.  At K = 256 -- the S x S projections of
// SGRAF's graph-reasoning steps (Fusionmodule.py:589-597), two per step over ~5 M node rows -- a tile is 8 chunks long and
// that kernel reaches 79-91 TFLOP/s; even at K = 2048 the fill / drain and its compiler-scheduled loop leave it at 132.  Here a workgroup owns ONE 128-column tile
// and streams down a contiguous range of 128-row tiles: operand chunks of tile t+1 are requested while the last chunks of
// tile t are multiplied, and tile t is flushed (bias, relu, stores) from its own accumulator set behind the MFMAs that fill the
// other set with tile t+1: the loop never drains and the stores never arrive in a burst.  The B panel of the
// workgroup (128 x K) is re-read from L2 for every row tile (it is the same 128 KB every time).
//
// The body is one generated asm statement (gemm_stream_asm.inc, tools/gen_gemm_stream.py) with hand-allocated registers:
// see scan_mainloop.inc for why the loops that hide loads from hipcc are not written in C++ any more.
// Preconditions (checked by the host wrapper): M % 128 == 0, N % 128 == 0, K % 64 == 0, K >= 128, 16-byte aligned rows.
#include "itr_common.h"
#include <mutex>

namespace itr {

constexpr int GS_BM = 128, GS_THREADS = 256, GS_NPLANE = 8;

struct GemmStreamArgs {
    const float *A, *B, *bias;
    float *C;
    int64_t lda, ldb, ldc;
    int64_t tiles_m, tiles_n;
    int K;
    int xcd_map;      // 1: the workgroups of one row range (all column tiles) sit on ONE XCD and share the A chunks in its L2
};

template <int ACT>      // epilogue: 0 none, 1 relu, 4 gelu (apply_act's codes)
__global__ __launch_bounds__(GS_THREADS, 2) void gemm_nt_stream_kernel(GemmStreamArgs g) {
    __shared__ float4 lds[2][2][GS_NPLANE][GS_BM];      // [buffer][operand][plane][row ^ plane]: 64 KB
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    // workgroup -> (column tile, contiguous range of row tiles).  Workgroups b, b + tiles_n, ... share a column tile.
    // (32-bit tile arithmetic, made provably wave-uniform for the "s" operands of the asm statement: hipcc expands a
    // division by a run-time value on the vector ALU)
    const unsigned tiles_n = (unsigned)g.tiles_n, tiles_m = (unsigned)g.tiles_m;
    // Consecutive workgroup ids go round the 8 XCDs.  With xcd_map, slot s of XCD x is (column tile s % tiles_n, row range
    // (s / tiles_n) * 8 + x): the tiles_n workgroups that walk the same rows run side by side under one L2 and A leaves HBM once.
    const unsigned nw = __builtin_amdgcn_readfirstlane(gridDim.x / tiles_n);
    const unsigned slot = g.xcd_map ? blockIdx.x >> 3 : blockIdx.x;
    const unsigned tn_u = __builtin_amdgcn_readfirstlane(slot % tiles_n);
    const unsigned iw = __builtin_amdgcn_readfirstlane(g.xcd_map ? (slot / tiles_n) * 8u + (blockIdx.x & 7u) : slot / tiles_n);
    const unsigned q_ = __builtin_amdgcn_readfirstlane(tiles_m / nw), r_ = __builtin_amdgcn_readfirstlane(tiles_m % nw);
    const int64_t tn = tn_u;
    const int64_t t0 = iw < r_ ? (int64_t)iw * (q_ + 1) : (int64_t)r_ * (q_ + 1) + (int64_t)(iw - r_) * q_;
    const int ntile = __builtin_amdgcn_readfirstlane((int)(q_ + (iw < r_ ? 1u : 0u)));
    if (ntile == 0) return;
    const int ld_row = tid >> 3, ld_p = tid & 7;
    const int fi = lane & 31, fg = lane >> 5;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)&lds[0][0][0][0];
    // per-lane global byte offsets of the four 32-row passes of one operand chunk (tile-relative)
    const unsigned oa0 = (unsigned)ld_row * (unsigned)g.lda * 4u + ld_p * 16u, oa1 = oa0 + 32u * (unsigned)g.lda * 4u;
    const unsigned oa2 = oa0 + 64u * (unsigned)g.lda * 4u, oa3 = oa0 + 96u * (unsigned)g.lda * 4u;
    const unsigned ob0 = (unsigned)ld_row * (unsigned)g.ldb * 4u + ld_p * 16u, ob1 = ob0 + 32u * (unsigned)g.ldb * 4u;
    const unsigned ob2 = ob0 + 64u * (unsigned)g.ldb * 4u, ob3 = ob0 + 96u * (unsigned)g.ldb * 4u;
    const unsigned ls0 = lds0 + (unsigned)(ld_p * GS_BM + (ld_row ^ ld_p)) * 16u;
    // fragment read addresses: plane p = 2q + fg, row (fi ^ p) within the wave's 64-row half ((base + fi) ^ p == base + (fi ^ p))
    auto faddr = [&](int q, int base, int oper) -> unsigned {
        const int p = 2 * q + fg;
        return lds0 + (unsigned)oper * (GS_NPLANE * GS_BM * 16u) + (unsigned)(p * GS_BM + base + (fi ^ p)) * 16u;
    };
    const unsigned fa0 = faddr(0, wm * 64, 0), fa1 = faddr(1, wm * 64, 0), fa2 = faddr(2, wm * 64, 0), fa3 = faddr(3, wm * 64, 0);
    const unsigned fb0 = faddr(0, wn * 64, 1), fb1 = faddr(1, wn * 64, 1), fb2 = faddr(2, wn * 64, 1), fb3 = faddr(3, wn * 64, 1);
    const int64_t n0 = tn * GS_BM;
    const float bias0 = g.bias ? g.bias[n0 + wn * 64 + fi] : 0.f;
    const float bias1 = g.bias ? g.bias[n0 + wn * 64 + 32 + fi] : 0.f;
    const unsigned voffc = (unsigned)(((int64_t)(wm * 64 + 4 * fg) * g.ldc + wn * 64 + fi) * 4);
    const char *pa = reinterpret_cast<const char *>(g.A + t0 * GS_BM * g.lda);
    const char *pb = reinterpret_cast<const char *>(g.B + n0 * g.ldb);
    char *pc = reinterpret_cast<char *>(g.C + t0 * GS_BM * g.ldc + n0);
    const unsigned tstep = (unsigned)(GS_BM * g.lda * 4);      // one row tile down
    const unsigned kbytes = (unsigned)g.K * 4u, ldc4 = (unsigned)g.ldc * 4u;
    const int nk2 = g.K / 64;
#include "gemm_stream_asm.inc"
}

// Host side: true when the streaming kernel took the call (otherwise the caller falls back to the tile kernels).
// algo (itr_gemm_nt_algo): 0 = the selection rule below, 1 = never (the caller runs the tile kernel), 2 = streaming with the plain
// tile map wherever the shape admits it, 3 = streaming with the XCD-aware map wherever the shape admits it.
bool gemm_nt_stream(const float *A, int64_t lda, const float *B, int64_t ldb, const float *bias, float *C, int64_t ldc, int64_t M,
                    int64_t N, int64_t K, int act, hipStream_t st, int *rc, int algo) {
    *rc = ITR_OK;
    if (algo == 1) return false;
    // resident workgroups of THIS device: CUs x what the occupancy query admits per CU (2 on gfx950: 64 KB of LDS each), cached
    // per device under a lock (several host threads / devices may call at once)
    static std::mutex mu;
    static int64_t resident_of[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return false;
    int64_t resident;
    {
        std::lock_guard<std::mutex> lock(mu);
        if (!resident_of[dev]) {
            hipDeviceProp_t prop;
            int per_cu = 0;
            if (hipGetDeviceProperties(&prop, dev) != hipSuccess ||
                hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gemm_nt_stream_kernel<0>, GS_THREADS, 0) != hipSuccess || per_cu < 1)
                return false;
            resident_of[dev] = (int64_t)prop.multiProcessorCount * per_cu;
        }
        resident = resident_of[dev];
    }
    const bool off = ITR_EXP_ENV("ITR_GEMM_STREAM") && atoi(ITR_EXP_ENV("ITR_GEMM_STREAM")) == 0;
    // (measured: ahead of the tile kernel at every K -- 179 200 x 1 024 x 2 048: 132 -> 147 TFLOP/s, 800 000 x 2 304 x 768: 120 -> 135,
    // 265 000 x 256 x 256: 79 -> 104; tools/gemm_stream_check.py.  ITR_GEMM_STREAM_KMAX caps K for experiments.)
    const int64_t kmax = ITR_EXP_ENV("ITR_GEMM_STREAM_KMAX") ? atoll(ITR_EXP_ENV("ITR_GEMM_STREAM_KMAX")) : (1ll << 40);
    if (off || (act != 0 && act != 1 && act != 4) || K % 64 != 0 || K < 128 || K > kmax || N % GS_BM != 0 || M < GS_BM) return false;
    if ((lda % 4) || (ldb % 4) || (reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(B) & 15)) return false;
    if ((uint64_t)lda * 4u * GS_BM >= (1ull << 31) || (uint64_t)ldb * 4u * GS_BM >= (1ull << 31) || (uint64_t)ldc * 4u * GS_BM >= (1ull << 31)) return false;
    const int64_t tiles_m = M / GS_BM, tiles_n = N / GS_BM;      // whole row tiles; a remainder of rows goes to the tile kernel
    const int64_t min_rounds = algo >= 2 ? 0 : ITR_EXP_ENV("ITR_GEMM_STREAM_MINROUNDS") ? atoll(ITR_EXP_ENV("ITR_GEMM_STREAM_MINROUNDS")) : 2;
    // streaming pays when every workgroup gets at least two tiles (measured: 16 384 x 1 024 x 1 024 = 2 rounds 130 -> 142 TFLOP/s,
    // 12 800 x 2 304 x 768 = 3.5 rounds 114 -> 132; at one round 5 000 x 3 072 x 1 024 loses 104 -> 96)
    if (tiles_m * tiles_n < min_rounds * resident) return false;
    int64_t grid = resident / tiles_n * tiles_n;
    if (grid < tiles_n) grid = tiles_n;
    const bool xcd_off = algo == 2 || (ITR_EXP_ENV("ITR_GEMM_STREAM_XCD") && atoi(ITR_EXP_ENV("ITR_GEMM_STREAM_XCD")) == 0);
    // (measured: +2-3 % from 4 column tiles up -- 179 200 x 1 024 x 2 048: 142 -> 147 TFLOP/s; nothing at 2)
    const int xcd_map = (!xcd_off && tiles_n >= 4 && grid % 8 == 0 && (grid / 8) % tiles_n == 0) ? 1 : 0;
    GemmStreamArgs g{A, B, bias, C, lda, ldb, ldc, tiles_m, tiles_n, (int)K, xcd_map};
    if (act == 1)
        hipLaunchKernelGGL(gemm_nt_stream_kernel<1>, dim3((unsigned)grid), dim3(GS_THREADS), 0, st, g);
    else if (act == 4)
        hipLaunchKernelGGL(gemm_nt_stream_kernel<4>, dim3((unsigned)grid), dim3(GS_THREADS), 0, st, g);
    else
        hipLaunchKernelGGL(gemm_nt_stream_kernel<0>, dim3((unsigned)grid), dim3(GS_THREADS), 0, st, g);
    if (hipGetLastError() != hipSuccess) { *rc = ITR_ERR_HIP; set_error("gemm_nt_stream: launch failed"); return true; }
    return true;
}

}  // namespace itr
