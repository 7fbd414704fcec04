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
    const unsigned long long t_loop0 = g.trace ? __builtin_amdgcn_s_memtime() : 0ull;
    // The D loop: one generated asm statement (tools/gen_sgraf_loc.py has the schedule and the register map).
#include "sgraf_loc_asm.inc"
    const unsigned long long t_loop1 = g.trace ? __builtin_amdgcn_s_memtime() : 0ull;

    // ---- epilogue: + bias, l2norm over the 256 features of a row (utils.py:10-15, eps 1e-8), store.
    // acc[i][j][r]: row = i*32 + (r & 3) + 8 (r >> 2) + 4 fg, column = wave*64 + j*32 + fi -- one column per lane, so storing from
    // the accumulators means 4-byte stores (256 per workgroup and wave...) and a cross-wave exchange for the row norms: tools/
    // loc_trace.py measured 48 000 cycles per workgroup for that form (15 % of a workgroup's life, during which the CU has ONE
    // workgroup in the D loop).  The tile goes through the idle operand buffers instead: [64 rows][288] floats (row stride
    // = 8 mod 16 sixteen-byte slots: the ds_read_b128 lane groups below hit 16 distinct slots), then 8 lanes own a row segment each:
    // row sum by three lane exchanges, one precise sqrt + divide per row, 16-byte stores of whole 128-byte lines.
    constexpr int XS_LD = 288;
    static_assert(LM * XS_LD * 4 <= (int)sizeof(LocSmem), "the output tile fits the operand buffers");
    __syncthreads();                                    // every wave is done with the operand buffers
    float *xs = reinterpret_cast<float *>(smem_raw);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float bv = g.bias[wave * 64 + j * 32 + fi];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                xs[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fg) * XS_LD + wave * 64 + j * 32 + fi] = acc[i][j][r] + bv;
    }
    __syncthreads();
    {
        const int seg = tid & 7;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int row = pass * 32 + (tid >> 3);
            float4 v[8];
            float ssq = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                v[k] = *reinterpret_cast<const float4 *>(xs + row * XS_LD + 32 * k + 4 * seg);
                ssq += v[k].x * v[k].x + v[k].y * v[k].y + v[k].z * v[k].z + v[k].w * v[k].w;
            }
            ssq += __shfl_xor(ssq, 1, 64);
            ssq += __shfl_xor(ssq, 2, 64);
            ssq += __shfl_xor(ssq, 4, 64);
            const float rn = 1.f / (sqrtf(ssq) + 1e-8f);
            float *out = g.X + (row0 + row) * LN + 4 * seg;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                *reinterpret_cast<float4 *>(out + 32 * k) = float4{v[k].x * rn, v[k].y * rn, v[k].z * rn, v[k].w * rn};
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

int sgraf_loc_fused(const float *P, const float *cn, const float *img, const float *wtiled, const float *W, const float *bias,
                    float *X, int64_t nb, int64_t n_tiles, int D, hipStream_t st) {
    static_assert(sizeof(LocSmem) == 80 * 1024, "two workgroups per CU");
    static bool attr_done = false;   // idempotent: racing callers set the same value
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(sgraf_loc_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)sizeof(LocSmem));
        if (e != hipSuccess) {
            set_error("sgraf_loc: cannot reserve %zu B of LDS: %s", sizeof(LocSmem), hipGetErrorString(e));
            return ITR_ERR_HIP;
        }
        attr_done = true;
    }
    if (nb == 0 || n_tiles == 0) return ITR_OK;
    LocArgs g{P, cn, img, wtiled, W, bias, X, nb, n_tiles, D, nullptr};
    const int64_t grid = ceil_div(n_tiles, (int64_t)8) * 8 * nb;
    // Debug only (tools/loc_trace.py): ITR_LOC_TRACE=<file> makes every launch synchronous and rewrites <file> with one record
    // per workgroup (hardware id + four s_memtime stamps), from which the tool rebuilds each CU's timeline.
    static const char *trace_path = getenv("ITR_LOC_TRACE");
    if (trace_path && *trace_path) {
        const size_t bytes = (size_t)grid * 5 * sizeof(unsigned long long);
        ITR_CHECK_HIP(hipMalloc(reinterpret_cast<void **>(&g.trace), bytes));
        ITR_CHECK_HIP(hipMemsetAsync(g.trace, 0, bytes, st));
        hipLaunchKernelGGL(sgraf_loc_kernel, dim3((unsigned)grid), dim3(256), sizeof(LocSmem), st, g);
        ITR_CHECK_LAUNCH("sgraf_loc");
        ITR_CHECK_HIP(hipStreamSynchronize(st));
        std::vector<unsigned long long> host((size_t)grid * 5);
        ITR_CHECK_HIP(hipMemcpy(host.data(), g.trace, bytes, hipMemcpyDeviceToHost));
        ITR_CHECK_HIP(hipFree(g.trace));
        if (FILE *f = fopen(trace_path, "wb")) { fwrite(host.data(), 1, bytes, f); fclose(f); }
        return ITR_OK;
    }
    hipLaunchKernelGGL(sgraf_loc_kernel, dim3((unsigned)grid), dim3(256), sizeof(LocSmem), st, g);
    ITR_CHECK_LAUNCH("sgraf_loc");
    return ITR_OK;
}

}  // namespace itr
