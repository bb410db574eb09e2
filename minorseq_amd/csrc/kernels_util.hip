// kernels_util.hip — device-side producers of the resident column-packed matrix:
//   synth_kernel      synthetic aligned CCS reads (jl_synth.h), one dword (8 reads) of one column per step
//   pack_rows_kernel  by-row uint8 codes -> column-packed nibbles (the first step of SURVEY §8 f1)
#include "jl_internal.h"

namespace {

constexpr int kSynthColsPerBlock = 64;

__global__ __launch_bounds__(256) void synth_kernel(jl_synth_plan pl, const uint8_t *__restrict__ ref,
                                                     uint8_t *__restrict__ msa, uint64_t col_stride,
                                                     uint64_t n_reads)
{
    const uint64_t t = (uint64_t)blockIdx.x * 256u + threadIdx.x;  // dword index in a column
    if (t * 4u >= col_stride) return;
    uint32_t hap[8], st[8], en[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const uint64_t i = t * 8u + r;
        if (i < n_reads) jl_synth_read(&pl, i, &hap[r], &st[r], &en[r]);
        else { hap[r] = 0; st[r] = 1; en[r] = 0; }  // empty range: padding reads are uncovered
    }
    const uint32_t c0 = blockIdx.y * kSynthColsPerBlock;
    const uint32_t c1 = min(pl.n_cols, c0 + kSynthColsPerBlock);
    for (uint32_t c = c0; c < c1; ++c) {
        const uint32_t rb = ref[c];
        uint32_t w = 0;
#pragma unroll
        for (int r = 0; r < 8; ++r) w |= jl_synth_cell(&pl, t * 8u + r, c, hap[r], st[r], en[r], rb) << (4 * r);
        *reinterpret_cast<uint32_t *>(msa + (uint64_t)c * col_stride + t * 4u) = w;
    }
}

// thread (x = column, y = dword of the column): gathers 8 reads of one column
__global__ __launch_bounds__(256) void pack_rows_kernel(const uint8_t *__restrict__ rows, uint64_t n_reads,
                                                         uint32_t n_cols, uint8_t *__restrict__ msa,
                                                         uint64_t col_stride)
{
    const uint32_t c = blockIdx.x * 64u + (threadIdx.x & 63u);
    const uint64_t t = (uint64_t)blockIdx.y * 4u + (threadIdx.x >> 6);
    if (c >= n_cols || t * 4u >= col_stride) return;
    uint32_t w = 0;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const uint64_t i = t * 8u + r;
        const uint32_t s = i < n_reads ? rows[i * n_cols + c] : 6u;
        w |= (s & 7u) << (4 * r);
    }
    *reinterpret_cast<uint32_t *>(msa + (uint64_t)c * col_stride + t * 4u) = w;
}

}  // namespace

void jl_launch_synth(jl_ctx *ctx, const jl_synth_plan *plan, const uint8_t *d_ref)
{
    const uint32_t n_dwords = (uint32_t)(ctx->col_stride / 4u);
    dim3 grid((n_dwords + 255u) / 256u, (ctx->n_cols + kSynthColsPerBlock - 1) / kSynthColsPerBlock);
    hipLaunchKernelGGL(synth_kernel, grid, dim3(256), 0, ctx->stream, *plan, d_ref, ctx->d_msa, ctx->col_stride,
                       ctx->n_reads);
}

void jl_launch_pack_rows(jl_ctx *ctx, const uint8_t *d_rows)
{
    const uint32_t n_dwords = (uint32_t)(ctx->col_stride / 4u);
    dim3 grid((ctx->n_cols + 63u) / 64u, (n_dwords + 3u) / 4u);
    hipLaunchKernelGGL(pack_rows_kernel, grid, dim3(256), 0, ctx->stream, d_rows, ctx->n_reads, ctx->n_cols,
                       ctx->d_msa, ctx->col_stride);
}
