// SGRAF local similarity nodes, fused (EncoderSimilarity.forward, Fusionmodule.py:425-427):
//
//     Context_img = l2norm(P V)                       SCAN_attention's weighted context (:657-662)
//     sim_loc     = l2norm(W_loc (Context_img - cap_i)^2 + b_loc)
//
// One workgroup = one (image, 64-column caption tile) = 64 node rows, all S = 256 output features.
// The D-long squared-difference rows are never written to HBM: per 32-wide slice of D a wave
//   stage 1:  ctx[16 rows x 32] = P'[16 x 36] V[36 x 32]      (v_mfma_f32_16x16x4_f32, 18 per slice; P' = P / (||ctx|| + eps)
//             lives in registers for the whole kernel)
//             a = (ctx - E)^2  ->  LDS, directly in the A-operand layout of stage 2
//   stage 2:  acc[64 x 64 per wave] += a[64 x 32] W_loc[256 x 32]^T   (v_mfma_f32_32x32x2_f32, 64 per slice)
// so the matrix core spends 1 / 8 of its time on producing the operand, against an HBM round trip of 2 x 4 D bytes per
// node row before (270 MB written and read back per image at COCO size).  The epilogue adds the bias, l2-normalises the
// 256-wide rows (cross-wave reduction through LDS) and stores the nodes.
//
// LDS: A slice 2 x 8 KB + W_loc slice 2 x 32 KB = 80 KB -> two workgroups per CU.  Operand layout as in gemm_f32.hip:
// 8 planes of float4 (plane p = k / 4), physical row = row ^ p (conflict-free ds_write_b128 / ds_read_b128); the stage-1
// results are scattered with ds_write_b32 whose 32 lanes per cycle cover 32 distinct banks (4 planes x 8 rows).
//
// The D loop is one generated asm statement (sgraf_loc_asm.inc, tools/gen_sgraf_loc.py): slice k+1 is PRODUCED (stage-1
// MFMAs, squared difference, LDS stores, refill of the register stage with slice k+2) behind the first half of slice k's
// stage-2 MFMAs, one barrier mid-way, the next fragments are read behind the second half.  The C++ loop it replaces ran
// stage 1 after stage 2 and then met a barrier: 127 -> 137 TFLOP/s (stage 1 + 2 flop, SGRAF 1k x 5k).
#include "scan_common.h"
#include <stdlib.h>
#include <vector>

namespace itr {

constexpr int LM = 64, LN = 256, LK = 32, LPL = LK / 4;

struct LocArgs {
    const float *P;        // [nb][ncols][36]   attention weights (SCAN emit)
    const float *cn;       // [nb][ncols]       1 / (||ctx|| + eps)
    const float *img;      // [nb][36][D]       region embeddings of this image block
    const float *wtiled;   // [ncols][D]        tile-packed word embeddings
    const float *W;        // [256][D]          sim_tranloc_w.weight
    const float *bias;     // [256]
    float *X;              // [nb][ncols][256]  out: l2-normalised local nodes
    int64_t nb, n_tiles;
    int D;
    unsigned long long *trace;   // debug (ITR_LOC_TRACE): [grid][5] = hardware id, s_memtime at entry / loop entry / loop exit / end
};

struct LocSmem {
    float4 a[2][LPL][LM];   // 16 KB
    float4 w[2][LPL][LN];   // 64 KB
};

__global__ __launch_bounds__(256, 2) void sgraf_loc_kernel(LocArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    LocSmem &sm = *reinterpret_cast<LocSmem *>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // XCD x (blockIdx % 8) owns the caption tiles ct = x (mod 8) and walks the images of the block for each of them,
    // so a tile of word embeddings is pulled into one L2 only; W_loc (1 MB) is resident in every L2.
    const int64_t bid = blockIdx.x;
    const int64_t xcd = bid & 7, idx = bid >> 3;
    const int64_t ct = (idx / g.nb) * 8 + xcd, ii = idx % g.nb;
    if (ct >= g.n_tiles) return;
    const unsigned long long t_entry = g.trace ? __builtin_amdgcn_s_memtime() : 0ull;
    const int64_t ncols = g.n_tiles * SC_NT;
    const int64_t row0 = ii * ncols + ct * SC_NT;
    const int D = g.D;
    const int nk = __builtin_amdgcn_readfirstlane(D / LK);

    // ---- stage-1 operands
    const int si = lane & 15, sg = lane >> 4;
    float pa[9];
    {
        const int64_t r = row0 + 16 * wave + si;
        const float sc = g.cn[r];
#pragma unroll
        for (int q = 0; q < 9; ++q) pa[q] = g.P[r * SC_R + 4 * q + sg] * sc;
    }
    // Loop-invariant addressing (the fp32 MFMA shares the vector ALU: the compiler's per-load 64-bit address arithmetic was
    // 225 VALU instructions per slice, ~19 % of the matrix time): uniform 64-bit bases advanced by scalar adds + ONE fixed
    // 32-bit per-lane byte offset per operand, loads issued through inline asm with our own s_waitcnt.
    const char *vbase = reinterpret_cast<const char *>(g.img + ii * SC_R * (int64_t)D);                       // V[k][d]
    const char *zbase = reinterpret_cast<const char *>(g.wtiled + (ct * SC_NT + 16 * wave) * (int64_t)D);    // E rows of this wave
    const char *wbase = reinterpret_cast<const char *>(g.W);
    const unsigned rowb = (unsigned)D * 4u;
    const unsigned voff_v = (unsigned)sg * rowb + si * 4u;            // V[4q + sg][d0 + si]      (q via the base, +16 columns via offset:64)
    const unsigned voff_z = 4u * sg * rowb + si * 4u;                 // E[4 sg + j][d0 + si]     (j via the base)
    // ---- stage-2 operands
    const int ld_row = tid >> 3, ld_p = tid & 7;
    const unsigned voff_w = (unsigned)ld_row * rowb + ld_p * 16u;     // W[ld_row + 32 s][kc*32 + 4 ld_p ..]  (s via the base)
    const int fi = lane & 31, fg = lane >> 5;
    // ---- LDS byte addresses (buffer 0; the other buffer and the passes are immediates in the generated loop)
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem_raw;
    unsigned ast[2][4];            // stage-1 result (row 16 wave + 4 sg + j, column k = 16 nt + si) in the A-operand layout
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = nt * 16 + si, p = k >> 2, row = 16 * wave + 4 * sg + j;
            ast[nt][j] = lds0 + (unsigned)((p * LM + (row ^ p)) * 16 + (k & 3) * 4);
        }
    const unsigned wst = lds0 + (unsigned)sizeof(sm.a) + (unsigned)(ld_p * LN + (ld_row ^ ld_p)) * 16u;      // + 512 s
    unsigned fa[4], fb[4];         // fragment reads: plane p = 2q + fg, physical row = row ^ p = base + (fi ^ p)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int p = 2 * q + fg;
        fa[q] = lds0 + (unsigned)(p * LM + (fi ^ p)) * 16u;                                                    // + 512 i
        fb[q] = lds0 + (unsigned)sizeof(sm.a) + (unsigned)(p * LN + wave * 64 + (fi ^ p)) * 16u;               // + 512 j
    }
    f32x16 acc[2][2];
    const float *bias_lane = g.bias + wave * 64 + 4 * fg;      // the accumulators start at the bias (generated prologue)
    const unsigned long long t_loop0 = g.trace ? __builtin_amdgcn_s_memtime() : 0ull;
    // The D loop: one generated asm statement (tools/gen_sgraf_loc.py has the schedule and the register map).
#include "sgraf_loc_asm.inc"
    const unsigned long long t_loop1 = g.trace ? __builtin_amdgcn_s_memtime() : 0ull;

    // ---- epilogue: l2norm over the 256 features of a row (utils.py:10-15, eps 1e-8), store.
    // The generated loop multiplies with swapped operands and starts from the bias, so acc[i][j] is the TRANSPOSED tile plus bias:
    // element r = feature wave*64 + 32 j + 8 (r >> 2) + 4 fg + (r & 3) of node row 32 i + fi -- a lane owns one row per i and runs
    // of four consecutive features.  What an instruction of this epilogue costs is its issue slot next to the co-resident
    // workgroup's MFMA stream (~55 cycles each, tools/loc_trace.py: 48 000 cycles per workgroup for the first form, which stored
    // 4-byte columns and exchanged all row sums through LDS; 25 000 for a version that transposed the tile through LDS), so the
    // count is what is minimised: packed squares and packed scaling (v_pk_fma_f32 / v_pk_mul_f32 on register pairs), one
    // v_permlane32_swap per row sum, 4 floats per lane through LDS for the cross-wave sums, sixteen 16-byte stores per lane.
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 sq[2] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2 v{acc[i][j][r], acc[i][j][r + 1]};
                sq[i] += v * v;
            }
    float ssum[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float s = sq[i][0] + sq[i][1];
        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);    // + the other half-wave (fg)
        ssum[i] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
    __syncthreads();                                    // every wave is done with the operand buffers
    float *part = reinterpret_cast<float *>(smem_raw);  // [64 rows][4 waves]
    if (fg == 0) {
        part[(fi) * 4 + wave] = ssum[0];
        part[(32 + fi) * 4 + wave] = ssum[1];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float4 ps = *reinterpret_cast<const float4 *>(part + (i * 32 + fi) * 4);
        const float rn = 1.f / (sqrtf((ps.x + ps.y) + (ps.z + ps.w)) + 1e-8f);
        const f32x2 rn2{rn, rn};
        float *out = g.X + (row0 + i * 32 + fi) * LN + wave * 64 + 4 * fg;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int b4 = 0; b4 < 4; ++b4) {
                const f32x2 lo = f32x2{acc[i][j][4 * b4], acc[i][j][4 * b4 + 1]} * rn2;
                const f32x2 hi = f32x2{acc[i][j][4 * b4 + 2], acc[i][j][4 * b4 + 3]} * rn2;
                *reinterpret_cast<float4 *>(out + 32 * j + 8 * b4) = float4{lo[0], lo[1], hi[0], hi[1]};
            }
    }
    if (g.trace && tid == 0) {
        // HW_REG_HW_ID (id 4): cu_id [11:8], sh_id [12], se_id [15:13]; HW_REG_XCC_ID (id 20): xcc_id [3:0]
        const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
        const unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the row stores are part of the workgroup's life
        unsigned long long *t = g.trace + (size_t)blockIdx.x * 5;
        t[0] = ((unsigned long long)xcc << 32) | hw;
        t[1] = t_entry; t[2] = t_loop0; t[3] = t_loop1; t[4] = __builtin_amdgcn_s_memtime();
    }
}

int allow_dynamic_lds(const void *kernel, size_t bytes);      // scan_train.hip

int sgraf_loc_fused(const float *P, const float *cn, const float *img, const float *wtiled, const float *W, const float *bias,
                    float *X, int64_t nb, int64_t n_tiles, int D, hipStream_t st) {
    static_assert(sizeof(LocSmem) == 80 * 1024, "two workgroups per CU");
    {   // once per (kernel, device), under a mutex (scan_train.hip): a process may drive several devices
        const int rc = allow_dynamic_lds(reinterpret_cast<const void *>(sgraf_loc_kernel), sizeof(LocSmem));
        if (rc != ITR_OK) return rc;
    }
    if (nb == 0 || n_tiles == 0) return ITR_OK;
    LocArgs g{P, cn, img, wtiled, W, bias, X, nb, n_tiles, D, nullptr};
    const int64_t grid = ceil_div(n_tiles, (int64_t)8) * 8 * nb;
    // Debug only (tools/loc_trace.py): ITR_LOC_TRACE=<file> makes every launch synchronous and rewrites <file> with one record
    // per workgroup (hardware id + four s_memtime stamps), from which the tool rebuilds each CU's timeline.
    static const char *trace_path = ITR_EXP_ENV("ITR_LOC_TRACE");
    // (experiment: ITR_LOC_ONE_PER_CU=1 asks for 82 KB of LDS, so only one workgroup fits a CU -- the loop's speed without a neighbour)
    static const size_t lds_bytes = (ITR_EXP_ENV("ITR_LOC_ONE_PER_CU") && atoi(ITR_EXP_ENV("ITR_LOC_ONE_PER_CU"))) ? 82 * 1024 : sizeof(LocSmem);
    if (lds_bytes != sizeof(LocSmem)) {
        static bool big_done = false;
        if (!big_done) {
            ITR_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(sgraf_loc_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
            big_done = true;
        }
    }
    if (trace_path && *trace_path) {
        const size_t bytes = (size_t)grid * 5 * sizeof(unsigned long long);
        ITR_CHECK_HIP(hipMalloc(reinterpret_cast<void **>(&g.trace), bytes));
        ITR_CHECK_HIP(hipMemsetAsync(g.trace, 0, bytes, st));
        hipLaunchKernelGGL(sgraf_loc_kernel, dim3((unsigned)grid), dim3(256), lds_bytes, st, g);
        ITR_CHECK_LAUNCH("sgraf_loc");
        ITR_CHECK_HIP(hipStreamSynchronize(st));
        std::vector<unsigned long long> host((size_t)grid * 5);
        ITR_CHECK_HIP(hipMemcpy(host.data(), g.trace, bytes, hipMemcpyDeviceToHost));
        ITR_CHECK_HIP(hipFree(g.trace));
        if (FILE *f = fopen(trace_path, "wb")) { fwrite(host.data(), 1, bytes, f); fclose(f); }
        return ITR_OK;
    }
    hipLaunchKernelGGL(sgraf_loc_kernel, dim3((unsigned)grid), dim3(256), lds_bytes, st, g);
    ITR_CHECK_LAUNCH("sgraf_loc");
    return ITR_OK;
}

}  // namespace itr
