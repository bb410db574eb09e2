// kernels_xwin.hip — device side of cross-window phasing with the reads sharded (SURVEY §8e option A):
//   xwin_pack_kernel    slice s of every variant column this rank owns (nine plane rows per position) -> one packed message
//                       per destination rank (the own slice straight into the compact matrix); also writes the phasing plan
//                       of the compact matrix (position k = columns 3k..3k+2), so no variant table travels to the device
//   xwin_assign_kernel  per-read ids of the slice from the merge's answer (haplotype of each exported group), with the
//                       completion word behind the last workgroup's ids
//   xwin_fetch_kernel   an all-gathered block from HBM into pinned host memory + completion word (one launch where a
//                       copy and a marker would be two stream operations and a host synchronisation)
// All three stream bytes: 16 B per lane, one wave = 1 KiB contiguous.
#include "jl_internal.h"
#include "result_pack.h"

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void plan_init(const jl_xw_pack_args &a)
{
    // the plan of a compact matrix is implicit: position k occupies columns 3k .. 3k + 2
    for (uint32_t k = threadIdx.x; k < a.vp_total; k += blockDim.x) {
        a.vpcols[k] = 3u * k;
        a.col2pos[3u * k] = k;
    }
    if (threadIdx.x == 0) {
        jl_phase_meta *m = a.meta;
        m->n_var = a.n_var;
        m->vp = a.vp_total;
        m->kwords = a.kwords;
        m->n_occupied = 0;
        m->overflow = 0;
        m->vp_true = a.vp_total;
        m->id_bits = 16;
        jl_phase_summary z = {0, 0, 0, 0, 0, 0, a.vp_total, 0};
        m->summary = z;
    }
}

// grid: x = 16 KiB pieces of a destination plane row (four 16-byte pieces per lane, loaded before any is stored), y = source
// plane row (9 per owned position: three columns x three planes), z = destination.  Bytes past the slice's last read are
// padding = code 6 = bit 0 clear, bits 1 and 2 set: 0x00 in plane 0, 0xFF in planes 1 and 2.
constexpr uint32_t kPackPieces = 4;
__global__ __launch_bounds__(256) void xwin_pack_kernel(jl_xw_pack_args a)
{
    if (a.meta && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) plan_init(a);
    if (blockIdx.z >= a.n_dst || blockIdx.y >= 9u * a.n_pos) return;
    const uint64_t dst_stride = a.d[blockIdx.z].dst_stride;
    const uint64_t bytes = a.d[blockIdx.z].bytes;
    const uint32_t tail_mask = a.d[blockIdx.z].tail_mask;   // bits of the last byte that are reads of the slice
    const uint32_t p = blockIdx.y / 9u, j = blockIdx.y - 9u * p;
    const uint32_t pad = (j % 3u) == 0u ? 0u : 0xFFu;
    const uint32_t padw = pad * 0x01010101u;
    const uint8_t *src = a.src[p] + (uint64_t)j * a.src_stride + a.d[blockIdx.z].byte_begin;
    uint8_t *dst = a.d[blockIdx.z].dst + (uint64_t)blockIdx.y * dst_stride;
    u32x4 v[kPackPieces];
    uint64_t off[kPackPieces];
#pragma unroll
    for (uint32_t k = 0; k < kPackPieces; ++k) {
        off[k] = (((uint64_t)blockIdx.x * kPackPieces + k) * 256u + threadIdx.x) * 16u;
        // the source row continues past the slice (more reads, or the row's own padding): a whole 16-byte load
        // below `bytes` is always inside it
        if (off[k] < bytes) v[k] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(src + off[k]));
    }
#pragma unroll
    for (uint32_t k = 0; k < kPackPieces; ++k) {
        if (off[k] >= dst_stride) continue;
        uint32_t w[4] = {padw, padw, padw, padw};
        if (off[k] < bytes) {
            w[0] = v[k].x; w[1] = v[k].y; w[2] = v[k].z; w[3] = v[k].w;
            if (off[k] + 16u >= bytes) {   // the slice ends in this piece: what lies past its last read becomes padding
#pragma unroll
                for (uint32_t b = 0; b < 16u; ++b) {
                    const uint32_t sh = 8u * (b & 3u);
                    if (off[k] + b >= bytes) w[b >> 2] = (w[b >> 2] & ~(0xFFu << sh)) | (pad << sh);
                    else if (off[k] + b + 1u == bytes) w[b >> 2] = (w[b >> 2] & ~((0xFFu & ~tail_mask) << sh)) | ((pad & ~tail_mask & 0xFFu) << sh);
                }
            }
        }
        *reinterpret_cast<uint4 *>(dst + off[k]) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

__global__ __launch_bounds__(256) void xwin_plan_kernel(jl_xw_pack_args a) { plan_init(a); }

// Eight reads per lane: one flag word, two 16-byte loads of slots; the slot's entry of slot_hap is the GROUP the
// exporting selection gave it, the table maps groups to haplotypes.
template <bool BYVAL>
__device__ __forceinline__ void xw_assign_body(const jl_xw_assign_args &a, const uint16_t *tab)
{
    for (uint64_t t = (uint64_t)blockIdx.x * 256u + threadIdx.x; t < a.n_dwords; t += (uint64_t)gridDim.x * 256u) {
        uint16_t h[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) h[r] = JL_HAP_DAMAGED;
        if (a.phased) {
            const uint32_t f = a.flagw[t];
            const uint4 s0 = *reinterpret_cast<const uint4 *>(a.read_slot + t * 8u);
            const uint4 s1 = *reinterpret_cast<const uint4 *>(a.read_slot + t * 8u + 4u);
            const uint32_t slot[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
            for (int r = 0; r < 8; ++r)
                if (((f >> (4 * r)) & 15u) == 0) {   // clean reads only: their slot is valid
                    const uint32_t q = a.slot_hap[slot[r]];
                    h[r] = q < a.n_groups ? tab[q] : (uint16_t)JL_HAP_INSUFFICIENT;
                }
        }
        jl_store_ids(a.read_hap, t, h, a.bits);
    }
}

// `host_stores`: this launch wrote host memory (every workgroup releases it at system scope before it arrives, so that the
// word cannot overtake another die's stores); ids that stay in HBM need none of that — the next kernel on the stream sees them
template <bool HOST_STORES>
__device__ __forceinline__ void xw_arrive_and_signal(uint32_t *arrive, uint32_t *seq_dev, volatile uint32_t *seq_host)
{
    if (!seq_host) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        if (HOST_STORES) __threadfence_system();   // this workgroup's stores leave its die's L2 before it arrives
        // (the run counter rides along with the arrival: no round trip of its own in the last workgroup)
        const uint32_t seq_before = __hip_atomic_load(seq_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t prev = __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == gridDim.x - 1u) {
            __hip_atomic_store(arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            jl_signal_done_from(seq_before, seq_dev, seq_host);
        }
    }
}

__global__ __launch_bounds__(256) void xwin_assign_val_kernel(jl_xw_assign_args a, jl_xw_hap_table tab)
{
    __shared__ uint16_t s_tab[JL_XW_TAB_MAX];
    for (uint32_t q = threadIdx.x; q < a.n_groups; q += 256u) s_tab[q] = tab.h[q];
    __syncthreads();
    xw_assign_body<true>(a, s_tab);
    xw_arrive_and_signal<false>(a.arrive, a.seq_dev, a.seq_host);
}

__global__ __launch_bounds__(256) void xwin_assign_ptr_kernel(jl_xw_assign_args a, const uint16_t *__restrict__ tab)
{
    xw_assign_body<false>(a, tab);
    xw_arrive_and_signal<false>(a.arrive, a.seq_dev, a.seq_host);
}

__global__ __launch_bounds__(256) void xwin_fetch_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, uint64_t n16,
                                                          uint32_t *arrive, uint32_t *seq_dev, volatile uint32_t *seq_host)
{
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * 256u) dst[i] = src[i];
    xw_arrive_and_signal<true>(arrive, seq_dev, seq_host);
}

}  // namespace

void jl_launch_xw_pack(const jl_xw_pack_args *a, hipStream_t st)
{
    if (a->n_pos == 0 || a->n_dst == 0) {
        if (a->meta) hipLaunchKernelGGL(xwin_plan_kernel, dim3(1), dim3(256), 0, st, *a);
        return;
    }
    uint64_t max_stride = 0;
    for (uint32_t k = 0; k < a->n_dst; ++k) max_stride = a->d[k].dst_stride > max_stride ? a->d[k].dst_stride : max_stride;
    const uint32_t gx = (uint32_t)((max_stride + 16383u) / 16384u);
    hipLaunchKernelGGL(xwin_pack_kernel, dim3(gx, 9u * a->n_pos, a->n_dst), dim3(256), 0, st, *a);
}

void jl_launch_xw_assign(const jl_xw_assign_args *a, const uint16_t *host_tab, const uint16_t *d_tab, hipStream_t st)
{
    // ids that stay in HBM: one workgroup per 2048 reads, at most 2048 of them (they loop)
    uint64_t blocks = (a->n_dwords + 255u) / 256u;
    if (blocks > 2048u) blocks = 2048u;
    if (blocks == 0) blocks = 1;
    if (d_tab) {
        hipLaunchKernelGGL(xwin_assign_ptr_kernel, dim3((uint32_t)blocks), dim3(256), 0, st, *a, d_tab);
    } else {
        jl_xw_hap_table tab;
        for (uint32_t q = 0; q < a->n_groups && q < JL_XW_TAB_MAX; ++q) tab.h[q] = host_tab[q];
        hipLaunchKernelGGL(xwin_assign_val_kernel, dim3((uint32_t)blocks), dim3(256), 0, st, *a, tab);
    }
}

void jl_launch_xw_fetch(const void *d_src, void *h_dst, uint64_t bytes, uint32_t *arrive, uint32_t *seq_dev, volatile uint32_t *seq_host,
                        hipStream_t st)
{
    const uint64_t n16 = (bytes + 15u) / 16u;
    uint64_t blocks = (n16 + 1023u) / 1024u;   // four 16-byte pieces per lane
    if (blocks > 64u) blocks = 64u;            // host-bound stores: PCIe is saturated long before the chip is full
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(xwin_fetch_kernel, dim3((uint32_t)blocks), dim3(256), 0, st, (const uint4 *)d_src, (uint4 *)h_dst, n16, arrive,
                       seq_dev, seq_host);
}
