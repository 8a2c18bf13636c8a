// aew_ops.hip — the HBM-bound ops of the hot path (everything that is not a GEMM).
#include "aew_common.h"

// =============================================================================================
// table-driven strided copy / convert / reduce   (weight pack, gradient unpack, NCL <-> NLC)
// =============================================================================================
// interleave forms of k_copy_table (K = -tr_a taps; conv weight [..][c][k] <-> [..][k][c]): a thread moves W channels
// of all K taps - K (or 2K) 16-byte loads, K 16-byte stores, both sides consecutive across the wave, the permutation in
// registers.  (The element-wise forms issue one 4-byte access per element on one of the two sides.)
//   tr_b = 1  interleave    (gradient unpack):  src[.. + t*ss[2] + c]  ->  dst[.. + c*K + t]        dims[2] = K, W = 4
//   tr_b = 2  de-interleave (weight pack):      src[.. + c*K + t]      ->  dst[.. + t*ds[3] + c]    dims[3] = K, W = 4 | 8 (bf16)
template <int K>
__device__ __forceinline__ void copy_ilv(const aew_copy_rec_t& r, unsigned item) {
    const bool ilv = r.tr_b == 1;
    const bool w8 = !ilv && r.dst_dtype == AEW_BF16;
    const int W = w8 ? 8 : 4;
    const int nc = (ilv ? r.dims[3] : r.dims[2]) / W;
    const unsigned total = (unsigned)r.dims[0] * (unsigned)r.dims[1] * (unsigned)nc;
    if (item >= total) return;
    const unsigned q = item % (unsigned)nc, o = item / (unsigned)nc;
    const int i1 = (int)(o % (unsigned)r.dims[1]), i0 = (int)(o / (unsigned)r.dims[1]);
    const float* sp = reinterpret_cast<const float*>(r.src) + i0 * r.ss[0] + i1 * r.ss[1];
    const int64_t dbase = i0 * r.ds[0] + i1 * r.ds[1];
    if (ilv) {
        float v[K][4];
#pragma unroll
        for (int t = 0; t < K; ++t) {
            const float4 x = *reinterpret_cast<const float4*>(sp + t * r.ss[2] + 4 * q);
            v[t][0] = x.x * r.scale; v[t][1] = x.y * r.scale; v[t][2] = x.z * r.scale; v[t][3] = x.w * r.scale;
        }
        float* dp = reinterpret_cast<float*>(r.dst) + dbase + (int64_t)4 * q * K;
#pragma unroll
        for (int j = 0; j < K; ++j) {                           // element e = 4j + u of the run is (channel e / K, tap e % K)
            float o4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) o4[u] = v[(4 * j + u) % K][(4 * j + u) / K];
            *reinterpret_cast<float4*>(dp + 4 * j) = make_float4(o4[0], o4[1], o4[2], o4[3]);
        }
        return;
    }
    if (w8) {
        float v[8 * K];
        const float* s0 = sp + (int64_t)8 * q * K;
#pragma unroll
        for (int j = 0; j < 2 * K; ++j) {
            const float4 x = *reinterpret_cast<const float4*>(s0 + 4 * j);
            v[4 * j] = x.x * r.scale; v[4 * j + 1] = x.y * r.scale; v[4 * j + 2] = x.z * r.scale; v[4 * j + 3] = x.w * r.scale;
        }
#pragma unroll
        for (int t = 0; t < K; ++t)
            *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(r.dst) + dbase + t * r.ds[3] + 8 * q) =
                make_uint4(pack2_bf16(v[t], v[K + t]), pack2_bf16(v[2 * K + t], v[3 * K + t]),
                           pack2_bf16(v[4 * K + t], v[5 * K + t]), pack2_bf16(v[6 * K + t], v[7 * K + t]));
    } else {
        float v[4 * K];
        const float* s0 = sp + (int64_t)4 * q * K;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const float4 x = *reinterpret_cast<const float4*>(s0 + 4 * j);
            v[4 * j] = x.x * r.scale; v[4 * j + 1] = x.y * r.scale; v[4 * j + 2] = x.z * r.scale; v[4 * j + 3] = x.w * r.scale;
        }
#pragma unroll
        for (int t = 0; t < K; ++t)
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(r.dst) + dbase + t * r.ds[3] + 4 * q) =
                make_float4(v[t], v[K + t], v[2 * K + t], v[3 * K + t]);
    }
}

__global__ void k_copy_table(const aew_copy_table_t t) {
    const int rec_i = t.block_rec[blockIdx.x];
    const aew_copy_rec_t r = t.recs[rec_i];
    if (r.tr_a < 0) {                                           // interleave forms (copy_ilv)
        const unsigned item = (blockIdx.x - (unsigned)r.first_block) * 256u + threadIdx.x;
        switch (-r.tr_a) {
            case 2: copy_ilv<2>(r, item); break;
            case 3: copy_ilv<3>(r, item); break;
            case 4: copy_ilv<4>(r, item); break;
            default: break;
        }
        return;
    }
    if (r.tr_a > 0) {
        // tiled form: one tr_a x tr_b tile of the (dims[2], dims[3]) plane through LDS - loads run along dims[3] (the
        // source's contiguous dim), stores along dims[2] (the destination's).  The element-wise forms touch one cache
        // line per element on one of the two sides of a transposing record (the encoder's dgrad-layout pack ran at
        // 0.9 TB/s).
        __shared__ float tile[4160];
        const int TA = r.tr_a, TB = r.tr_b, pitch = TA | 1;
        const unsigned na = (unsigned)((r.dims[3] + TA - 1) / TA), nb = (unsigned)((r.dims[2] + TB - 1) / TB);
        unsigned q = blockIdx.x - (unsigned)r.first_block;
        const int a0 = (int)(q % na) * TA; q /= na;
        const int b0 = (int)(q % nb) * TB; q /= nb;
        const int i1 = (int)(q % (unsigned)r.dims[1]), i0 = (int)(q / (unsigned)r.dims[1]);
        const float* sp = reinterpret_cast<const float*>(r.src) + i0 * r.ss[0] + i1 * r.ss[1];
        const int64_t dbase = i0 * r.ds[0] + i1 * r.ds[1];
        const int n = TA * TB;
        // (four loads per thread issued before the first LDS write: a tile of up to 64 x 64 is 16 loads per thread, and what
        // bounds this kernel is the bytes it keeps in flight - 1024-element tiles ran at 2.2 TB/s)
        for (int e0 = threadIdx.x; e0 < n; e0 += 1024) {
            float v[4];
            int at[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = e0 + u * 256;
                const int bl = e / TA, al = e - bl * TA;
                const int a = a0 + al, b = b0 + bl;
                at[u] = e < n ? bl * pitch + al : -1;
                v[u] = (e < n && a < r.dims[3] && b < r.dims[2]) ? sp[b * r.ss[2] + a * r.ss[3]] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (at[u] >= 0) tile[at[u]] = v[u] * r.scale;
        }
        __syncthreads();
        if (r.dst_dtype == AEW_BF16) {
            uint16_t* dp = reinterpret_cast<uint16_t*>(r.dst) + dbase;
            // two consecutive dims[2] positions per thread (one 4-byte store) when every pair is aligned
            const bool pair = !((TB | r.dims[2]) & 1) && !((r.ds[3] | dbase) & 1) && !((uintptr_t)r.dst & 3);
            if (pair) {
                const int hb = TB >> 1;
                for (int e = threadIdx.x; e < TA * hb; e += 256) {
                    const int al = e / hb, bl = (e - al * hb) * 2;
                    const int a = a0 + al, b = b0 + bl;
                    if (a < r.dims[3] && b < r.dims[2])
                        *reinterpret_cast<uint32_t*>(dp + b + a * r.ds[3]) = pack2_bf16(tile[bl * pitch + al], tile[(bl + 1) * pitch + al]);
                }
            } else {
                for (int e = threadIdx.x; e < n; e += 256) {
                    const int al = e / TB, bl = e - al * TB;
                    const int a = a0 + al, b = b0 + bl;
                    if (a < r.dims[3] && b < r.dims[2]) dp[b + a * r.ds[3]] = f2bf(tile[bl * pitch + al]);
                }
            }
        } else {
            float* dp = reinterpret_cast<float*>(r.dst) + dbase;
            for (int e = threadIdx.x; e < n; e += 256) {
                const int al = e / TB, bl = e - al * TB;
                const int a = a0 + al, b = b0 + bl;
                if (a < r.dims[3] && b < r.dims[2]) dp[b + a * r.ds[3]] = tile[bl * pitch + al];
            }
        }
        return;
    }
    // vector form: 4 consecutive elements of the last dim per thread when the fp32 source is
    // contiguous there (the gradient-unpack records: float4 loads over every slab)
    // destination-vector form: fp32 -> bf16 pack records whose LAST dim is contiguous in the destination:
    // 8 (strided) fp32 loads, one 16-byte bf16 store (the element-wise form issued 2-byte scattered stores)
    const bool vecd = r.src_dtype == AEW_F32 && r.dst_dtype == AEW_BF16 && r.red_n == 1 && r.ds[3] == 1 &&
                      (r.dims[3] & 7) == 0 &&
                      (((uintptr_t)r.dst | (uintptr_t)(r.ds[2] * 2) | (uintptr_t)(r.ds[1] * 2) | (uintptr_t)(r.ds[0] * 2)) & 15) == 0;
    // same idea for fp32 -> fp32 permuting copies (encoder weight pack): 4 strided loads, one float4 store
    const bool vecf = r.src_dtype == AEW_F32 && r.dst_dtype == AEW_F32 && r.red_n == 1 && !r.accumulate &&
                      r.ds[3] == 1 && r.ss[3] != 1 && (r.dims[3] & 3) == 0 &&
                      (((uintptr_t)r.dst | (uintptr_t)(r.ds[2] * 4) | (uintptr_t)(r.ds[1] * 4) | (uintptr_t)(r.ds[0] * 4)) & 15) == 0;
    const bool vec = !vecd && !vecf && r.src_dtype == AEW_F32 && r.ss[3] == 1 && (r.dims[3] & 3) == 0 &&
                     (((uintptr_t)r.src | (uintptr_t)(r.ss[2] * 4) | (uintptr_t)(r.ss[1] * 4) |
                       (uintptr_t)(r.ss[0] * 4) | (uintptr_t)(r.red_stride * 4)) & 15) == 0;
    const int d3 = (vec || vecf) ? r.dims[3] >> 2 : (vecd ? r.dims[3] >> 3 : r.dims[3]);
    const int64_t total = (int64_t)r.dims[0] * r.dims[1] * r.dims[2] * d3;
    // the host sizes the grid as ceil(elements / 1024) blocks per record; the vector forms cover >= 4
    // elements per thread, so they take ONE item per thread (4x the threads in flight for the slab loads of
    // the unpack records) and the scalar form four
    const int per_thread = (vec || vecf || vecd) ? 1 : 4;
    const int64_t base = (int64_t)(blockIdx.x - r.first_block) * (256 * per_thread);
    for (int u = 0; u < per_thread; ++u) {
        const int64_t e64 = base + u * 256 + threadIdx.x;
        if (e64 >= total) return;
        // (32-bit index arithmetic: a record never has 2^32 items, and three 64-bit divisions per item were most of
        // what a pack / unpack thread executed)
        unsigned e = (unsigned)e64;
        const unsigned q3 = e / (unsigned)d3;
        const int i3 = (int)(e - q3 * (unsigned)d3) * ((vec || vecf) ? 4 : (vecd ? 8 : 1)); e = q3;
        const unsigned q2 = e / (unsigned)r.dims[2];
        const int i2 = (int)(e - q2 * (unsigned)r.dims[2]); e = q2;
        const unsigned q1 = e / (unsigned)r.dims[1];
        const int i1 = (int)(e - q1 * (unsigned)r.dims[1]);
        const int i0 = (int)q1;
        const int64_t so = i0 * r.ss[0] + i1 * r.ss[1] + i2 * r.ss[2] + i3 * r.ss[3];
        const int64_t dof = i0 * r.ds[0] + i1 * r.ds[1] + i2 * r.ds[2] + i3 * r.ds[3];
        if (vecf) {
            const float* sp = reinterpret_cast<const float*>(r.src) + so;
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(r.dst) + dof) =
                make_float4(sp[0] * r.scale, sp[r.ss[3]] * r.scale, sp[2 * r.ss[3]] * r.scale, sp[3 * r.ss[3]] * r.scale);
            continue;
        }
        if (vecd) {
            const float* sp = reinterpret_cast<const float*>(r.src) + so;
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = sp[k * r.ss[3]] * r.scale;
            *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(r.dst) + dof) =
                make_uint4(pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]), pack2_bf16(v[4], v[5]), pack2_bf16(v[6], v[7]));
            continue;
        }
        if (vec) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            const float* sp = reinterpret_cast<const float*>(r.src) + so;
            int q = 0;
            // 8 slab loads in flight per thread; the additions stay in slab order (deterministic)
            for (; q + 8 <= r.red_n; q += 8) {
                float4 v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const float4*>(sp + (int64_t)(q + k) * r.red_stride);
#pragma unroll
                for (int k = 0; k < 8; ++k) { acc.x += v[k].x; acc.y += v[k].y; acc.z += v[k].z; acc.w += v[k].w; }
            }
            for (; q < r.red_n; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(sp + (int64_t)q * r.red_stride);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
            const float a4[4] = {acc.x * r.scale, acc.y * r.scale, acc.z * r.scale, acc.w * r.scale};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int64_t d = dof + c * r.ds[3];
                if (r.dst_dtype == AEW_BF16) reinterpret_cast<uint16_t*>(r.dst)[d] = f2bf(a4[c]);
                else if (r.accumulate) reinterpret_cast<float*>(r.dst)[d] += a4[c];
                else reinterpret_cast<float*>(r.dst)[d] = a4[c];
            }
            continue;
        }
        float acc = 0.f;
        for (int q = 0; q < r.red_n; ++q) {
            const int64_t si = so + q * r.red_stride;
            acc += (r.src_dtype == AEW_BF16) ? bf2f(reinterpret_cast<const uint16_t*>(r.src)[si])
                                             : reinterpret_cast<const float*>(r.src)[si];
        }
        acc *= r.scale;
        if (r.dst_dtype == AEW_BF16) reinterpret_cast<uint16_t*>(r.dst)[dof] = f2bf(acc);
        else if (r.accumulate) reinterpret_cast<float*>(r.dst)[dof] += acc;
        else reinterpret_cast<float*>(r.dst)[dof] = acc;
    }
}

// =============================================================================================
// VQ: nearest code.  One wave per query; lane scans codes lane, lane+64, ...; the arithmetic is
// the exact order of oracle/exact_chain.c (fma chains, IEEE sqrt and divide, strict '<',
// lowest index wins).   vqema_bn.py:135-142, vq_bn.py:39-41
// =============================================================================================
// Scan of codes [k0, k1) by one block: per-code arithmetic in the exact order of oracle/exact_chain.c;
// returns the block's (min, lowest index) in every thread.
__device__ __forceinline__ void vq_scan(const aew_vq_nearest_t& p, const float* z, float zn, int k0, int k1,
                                        float& best, int& bi) {
    __shared__ float sh_d[4];
    __shared__ int sh_i[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    best = INFINITY;
    bi = 0x7fffffff;
    for (int k = k0 + threadIdx.x; k < k1; k += 256) {
        const float* c = p.emb + (int64_t)k * p.d;
        float dd = 0.f, qq = 0.f;
        if ((p.d & 3) == 0) {                        // 16-byte loads of the code row; arithmetic order unchanged
            for (int j = 0; j < p.d; j += 4) {
                const float4 c4 = *reinterpret_cast<const float4*>(c + j);
                const float cv[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float t = __fsub_rn(z[j + r], cv[r]);
                    dd = __fmaf_rn(t, t, dd);
                    qq = __fmaf_rn(cv[r], cv[r], qq);
                }
            }
        } else {
            for (int j = 0; j < p.d; ++j) {
                const float cj = c[j];
                const float t = __fsub_rn(z[j], cj);
                dd = __fmaf_rn(t, t, dd);
                qq = __fmaf_rn(cj, cj, qq);
            }
        }
        float v;
        if (p.metric == 0) v = __fdiv_rn(sqrtf(dd), __fadd_rn(zn, sqrtf(qq)));
        else v = dd;
        if (v < best) { best = v; bi = k; }          // ascending k within the thread: first min wins
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov < best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (lane == 0) { sh_d[wv] = best; sh_i[wv] = bi; }
    __syncthreads();
    best = sh_d[0]; bi = sh_i[0];
#pragma unroll
    for (int w = 1; w < 4; ++w)
        if (sh_d[w] < best || (sh_d[w] == best && sh_i[w] < bi)) { best = sh_d[w]; bi = sh_i[w]; }
}

__device__ __forceinline__ float vq_znorm(const aew_vq_nearest_t& p, const float* z) {
    float zz = 0.f;
    for (int j = 0; j < p.d; ++j) zz = __fmaf_rn(z[j], z[j], zz);
    return sqrtf(zz);                    // sqrtf is IEEE-rounded; __fsqrt_rn lowers to a bare v_sqrt_f32
}

__device__ __forceinline__ void vq_emit(const aew_vq_nearest_t& p, int q, float best, int bi) {
    if (bi == 0x7fffffff) bi = 0;                    // all-NaN row: torch.min would return NaN; pick 0
    if (threadIdx.x == 0) { p.ind[q] = bi; p.dist[q] = best; }
    float* zq = p.zq + (int64_t)q * p.d_pitch;
    for (int j = threadIdx.x; j < p.d_pitch; j += blockDim.x) zq[j] = j < p.d ? p.emb[(int64_t)bi * p.d + j] : 0.f;
}

__global__ __launch_bounds__(256) void k_vq_nearest(const aew_vq_nearest_t p) {
    // one block (4 waves) per query; wave w scans codes w*64+lane, +256, ...  The argmin with
    // lowest-index tie-break is independent of the scan order, so the result equals the
    // sequential oracle loop.
    const int q = blockIdx.x;
    const float* z = p.ze + (int64_t)q * p.d_pitch;
    float best; int bi;
    vq_scan(p, z, vq_znorm(p, z), 0, p.K, best, bi);
    vq_emit(p, q, best, bi);
}

// split form: grid (n_split, Q); block (s, q) scans its slice of the codebook and leaves (min, index) in
// scratch; k_vq_nearest_combine takes the minimum over the slices (ties: lowest index, i.e. lowest slice)
__global__ __launch_bounds__(256) void k_vq_nearest_part(const aew_vq_nearest_t p) {
    const int s = blockIdx.x, q = blockIdx.y;
    const int per = (p.K + p.n_split - 1) / p.n_split;
    const float* z = p.ze + (int64_t)q * p.d_pitch;
    float best; int bi;
    vq_scan(p, z, vq_znorm(p, z), s * per, min(p.K, (s + 1) * per), best, bi);
    if (threadIdx.x == 0) {
        float* pd = reinterpret_cast<float*>(p.scratch) + ((int64_t)q * p.n_split + s) * 2;
        pd[0] = best;
        reinterpret_cast<int*>(pd)[1] = bi;
    }
}
// The same partial scan for QB queries per block: a thread keeps ITS code row (D floats) and its squared norm in
// registers and meets QB encoder outputs staged in LDS - the row is fetched once instead of once per query (the fetch
// is one 128-byte row per lane, the expensive access of the scan).  Every (query, code) distance is the same ascending
// fma chain as in vq_scan and the minimum takes the lowest index on ties, so indices and distances are bit-identical.
template <int D, int QB>
__global__ __launch_bounds__(256) void k_vq_nearest_part_mq(const aew_vq_nearest_t p) {
    __shared__ float sh_z[QB][D];
    __shared__ float sh_zn[QB];
    __shared__ float sh_d[QB][4];
    __shared__ int sh_i[QB][4];
    const int s = blockIdx.x, q0 = blockIdx.y * QB;
    const int per = (p.K + p.n_split - 1) / p.n_split;           // <= 256 (launcher)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nq = min(QB, p.Q - q0);
    for (int e = threadIdx.x; e < QB * D; e += 256) {
        const int qi = e / D, j = e - qi * D;
        sh_z[qi][j] = qi < nq ? p.ze[(int64_t)(q0 + qi) * p.d_pitch + j] : 0.f;
    }
    __syncthreads();
    if (threadIdx.x < QB) {                                      // ||z||: the chain of vq_znorm
        float zz = 0.f;
        for (int j = 0; j < D; ++j) zz = __fmaf_rn(sh_z[threadIdx.x][j], sh_z[threadIdx.x][j], zz);
        sh_zn[threadIdx.x] = sqrtf(zz);
    }
    const int k = s * per + threadIdx.x;
    const bool live = threadIdx.x < per && k < p.K;
    float c[D];
    float qq = 0.f;
    {
        const float* cp = p.emb + (int64_t)(live ? k : 0) * D;
#pragma unroll
        for (int j = 0; j < D; j += 4) {
            const float4 c4 = *reinterpret_cast<const float4*>(cp + j);
            c[j] = c4.x; c[j + 1] = c4.y; c[j + 2] = c4.z; c[j + 3] = c4.w;
        }
#pragma unroll
        for (int j = 0; j < D; ++j) qq = __fmaf_rn(c[j], c[j], qq);
    }
    const float sq = sqrtf(qq);
    __syncthreads();
#pragma unroll 1
    for (int qi = 0; qi < nq; ++qi) {
        float dd = 0.f;
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const float t = __fsub_rn(sh_z[qi][j], c[j]);
            dd = __fmaf_rn(t, t, dd);
        }
        float best = INFINITY;
        int bi = 0x7fffffff;
        if (live) {
            const float v = p.metric == 0 ? __fdiv_rn(sqrtf(dd), __fadd_rn(sh_zn[qi], sq)) : dd;
            if (v < best) { best = v; bi = k; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov < best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        if (lane == 0) { sh_d[qi][wv] = best; sh_i[qi][wv] = bi; }
    }
    __syncthreads();
    if (threadIdx.x < nq) {
        const int qi = threadIdx.x;
        float best = sh_d[qi][0];
        int bi = sh_i[qi][0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (sh_d[qi][w] < best || (sh_d[qi][w] == best && sh_i[qi][w] < bi)) { best = sh_d[qi][w]; bi = sh_i[qi][w]; }
        float* pd = reinterpret_cast<float*>(p.scratch) + ((int64_t)(q0 + qi) * p.n_split + s) * 2;
        pd[0] = best;
        reinterpret_cast<int*>(pd)[1] = bi;
    }
}

__global__ __launch_bounds__(64) void k_vq_nearest_combine(const aew_vq_nearest_t p) {
    const int q = blockIdx.x;
    const float* pd = reinterpret_cast<const float*>(p.scratch) + (int64_t)q * p.n_split * 2;
    float best = INFINITY;
    int bi = 0x7fffffff;
    for (int s = 0; s < p.n_split; ++s) {
        const float v = pd[2 * s];
        const int i = reinterpret_cast<const int*>(pd)[2 * s + 1];
        if (v < best || (v == best && i < bi)) { best = v; bi = i; }
    }
    vq_emit(p, q, best, bi);
}

// z_sum / n_sum: one thread per (code, channel), queries in ascending order (deterministic and
// bit-identical to the oracle).  vqema_bn.py:172-188
__global__ __launch_bounds__(256) void k_vq_stats(const aew_vq_stats_t p) {
    // the code index of every query is staged through LDS in chunks (a global load per query and thread
    // made this loop a chain of L2 round trips: 47 us for Q = 232)
    __shared__ int sh_ind[1024];
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = e < (int64_t)p.K * p.d;
    const int k = live ? (int)(e / p.d) : -1, j = live ? (int)(e % p.d) : 0;
    float s = 0.f, n = 0.f;
    for (int q0 = 0; q0 < p.Q; q0 += 1024) {
        const int nq = min(1024, p.Q - q0);
        __syncthreads();
        for (int i = threadIdx.x; i < nq; i += blockDim.x) sh_ind[i] = (int)p.ind[q0 + i];
        __syncthreads();
        for (int i = 0; i < nq; ++i)                      // queries in ascending order: deterministic,
            if (sh_ind[i] == k) {                         // bit-identical to the oracle
                s = __fadd_rn(s, p.ze[(int64_t)(q0 + i) * p.d_pitch + j]);
                n = __fadd_rn(n, 1.0f);
            }
    }
    if (!live) return;
    p.z_sum[e] = s;
    if (j == 0) {
        p.n_sum[k] = n;
        if (p.hist) p.hist[k] += n;
    }
}

// Few queries (a training step: Q = 232 against K x d = 131 072 accumulators): the kernel above spends its time
// finding out that almost every code has no query.  Here the outputs are cleared first and one wave per query does
// the work of its code if it is the FIRST query that maps to it: the same ascending-q chain of fp32 adds starting
// from 0, so the results are bit-identical.  Q <= 1024 (the index list lives in LDS).
__global__ __launch_bounds__(256) void k_vq_stats_few(const aew_vq_stats_t p) {
    __shared__ int sh_ind[1024];
    for (int i = threadIdx.x; i < p.Q; i += blockDim.x) sh_ind[i] = (int)p.ind[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);           // one wave per query
    if (q >= p.Q) return;
    const int k = sh_ind[q];
    // 64 queries per ballot: matches before q mean another wave owns code k; the matches from q on are visited in
    // ascending order through the set bits of the masks
    const int nch = (p.Q + 63) >> 6;
    float s = 0.f, n = 0.f;                                       // lane j < d accumulates channel j (d <= 64)
    for (int c = 0; c < nch; ++c) {
        const int i = c * 64 + lane;
        unsigned long long m = __ballot(i < p.Q && sh_ind[i] == k);
        if (c * 64 < q) {
            const unsigned long long before = (q - c * 64 >= 64) ? ~0ull : ((1ull << (q - c * 64)) - 1ull);
            if (m & before) return;                               // an earlier query owns code k
            m &= ~before;
        }
        while (m) {                                               // 16 rows in flight, then added in order: a collapsed
            int qi[16], cnt = 0;                                  // codebook sends every query to one code (232 rows)
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (m) { qi[u] = c * 64 + __builtin_ctzll(m); m &= m - 1; cnt = u + 1; } else qi[u] = 0;
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = (u < cnt && lane < p.d) ? p.ze[(int64_t)qi[u] * p.d_pitch + lane] : 0.f;
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (u < cnt) { s = __fadd_rn(s, v[u]); n = __fadd_rn(n, 1.0f); }
        }
    }
    if (lane < p.d) p.z_sum[(int64_t)k * p.d + lane] = s;
    if (lane == 0) {
        p.n_sum[k] = n;
        if (p.hist) p.hist[k] += n;
    }
}

__global__ void k_vq_ema(const aew_vq_ema_t p) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)p.K * p.d) return;
    if (p.guard && *p.guard) return;
    const int k = (int)(e / p.d), j = (int)(e % p.d);
    const float nu = __fadd_rn(__fmul_rn(p.gamma, p.numer[e]), __fmul_rn(p.gamma_comp, p.z_sum[e]));
    const float de = __fadd_rn(__fmul_rn(p.gamma, p.denom[k]), __fmul_rn(p.gamma_comp, p.n_sum[k]));
    p.numer[e] = nu;
    // update_codebook 2 = k-means centroid step: a code that owns no sample keeps its position
    if (p.update_codebook == 1 || (p.update_codebook == 2 && de > 0.f)) p.emb[e] = __fdiv_rn(nu, de);
    // every thread of row k computes the same `de`; the write is deferred to a second kernel so
    // no thread reads denom[k] after another thread of the row has overwritten it
    (void)j;
}
__global__ void k_vq_ema_denom(const aew_vq_ema_t p) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= p.K) return;
    if (p.guard && *p.guard) return;
    p.denom[k] = __fadd_rn(__fmul_rn(p.gamma, p.denom[k]), __fmul_rn(p.gamma_comp, p.n_sum[k]));
}

// d(ze) = d(zq) [straight-through, vqema_bn.py:44-45] + coef * d(dist_min)/d(ze)
//   scaled_l2: dist = u/v, u = ||z-q||, v = ||z|| + ||q||
//       d dist/dz = (z-q)/(u v) - u z / (v^2 ||z||)
//   sq_l2:     d dist/dz = 2 (z-q);  VQ also gets d/d(emb) of the l2 term (vq_bn.py:78)
__global__ __launch_bounds__(256) void k_vq_bwd(const aew_vq_bwd_t p, int det) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= p.Q) return;
    const float* z = p.ze + (int64_t)q * p.d_pitch;
    const int64_t k = p.ind[q];
    const float* c = p.emb + k * p.d;
    float dd = 0.f, zz = 0.f, qq = 0.f;
    for (int j = lane; j < p.d; j += 64) {
        const float t = z[j] - c[j];
        dd += t * t; zz += z[j] * z[j]; qq += c[j] * c[j];
    }
    dd = wave_sum(dd); zz = wave_sum(zz); qq = wave_sum(qq);
    const float u = sqrtf(dd), zn = sqrtf(zz), v = zn + sqrtf(qq);
    const float gm = p.gmul ? p.gmul[0] : 1.0f;        // upstream gradient of this backward call
    const float coef = p.coef * gm, demb_coef = p.demb_coef * gm;
    for (int j = lane; j < p.d_pitch; j += 64) {
        float gr = 0.f;
        if (j < p.d) {
            const float t = z[j] - c[j];
            float dj;
            if (p.metric == 0) dj = (u > 0.f ? t / (u * v) : 0.f) - (zn > 0.f ? u * z[j] / (v * v * zn) : 0.f);
            else dj = 2.0f * t;
            gr = p.dzq[(int64_t)q * p.d_pitch + j] + coef * dj;
            if (p.demb && !det) atomicAdd(p.demb + k * p.d + j, demb_coef * (-2.0f * t));
        }
        p.dze[(int64_t)q * p.d_pitch + j] = gr;
    }
    if (p.demb && det) {
        // d/d(emb[k]) sums over the queries that chose code k.  Deterministic form: the wave of the FIRST such query adds
        // the terms of all of them in ascending query order and is the only writer of row k (pre-zeroed by the caller).
        bool first = true;
        for (int c0 = 0; c0 < q && first; c0 += 64) {
            const int i = c0 + lane;
            if (__any(i < q && p.ind[i] == k)) first = false;
        }
        if (!first) return;
        for (int j0 = 0; j0 < p.d; j0 += 64) {
            const int j = j0 + lane;
            float acc = 0.f;
            for (int c0 = q & ~63; c0 < p.Q; c0 += 64) {
                const int i = c0 + lane;
                unsigned long long m = __ballot(i >= q && i < p.Q && p.ind[i] == k);
                while (m) {
                    const int qi = c0 + __builtin_ctzll(m);
                    m &= m - 1;
                    if (j < p.d) acc += demb_coef * (-2.0f * (p.ze[(int64_t)qi * p.d_pitch + j] - c[j]));
                }
            }
            if (j < p.d) p.demb[k * p.d + j] += acc;
        }
    }
}

// =============================================================================================
// jitter gather / scatter   (wavenet.py:330-336)
// =============================================================================================
__global__ void k_lc_gather(const aew_lc_gather_t p) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)p.B * p.N * p.C_pad;
    if (e >= total) return;
    const int c = (int)(e % p.C_pad);
    const int t = (int)((e / p.C_pad) % p.N);
    const int b = (int)(e / ((int64_t)p.C_pad * p.N));
    float v = 0.f;
    if (c < p.C) {
        // The reference's Jitter emits t-1+x (x in 0..2) for every t < n (jitter.py:29-33), so the last entry may be n,
        // one past the end; and a jitter row made for the mel frames is longer than the encoder output it indexes here.
        // torch.gather would raise on such an index; the intended neighbour is the last row, so indices are clamped
        // (reads and the matching gradient scatter stay inside the source matrix for every input).
        int64_t j = p.jitter[(int64_t)b * p.jit_pitch + t];
        j = j < 0 ? 0 : (j > p.N - 1 ? p.N - 1 : j);
        if (p.take_compat) {
            // torch.take on the flattened (B, C, N) tensor with index b*N + j, expanded over c:
            // flat index -> (b', c', n') of the NCL tensor
            const int64_t flat = (int64_t)b * p.N + j;
            const int64_t bb = flat / ((int64_t)p.C * p.N), cc = (flat / p.N) % p.C, nn = flat % p.N;
            v = p.src[bb * p.src_bs + nn * p.src_pitch + cc];
        } else {
            v = p.src[(int64_t)b * p.src_bs + j * p.src_pitch + c];
        }
    }
    p.dst[(int64_t)b * p.dst_bs + (int64_t)t * p.dst_pitch + c] = f2bf(v);
}

// Gather form of the scatter (deterministic, no atomics, no pre-zeroed target): one thread per element of dsrc adds the
// rows t whose jitter index points at it, t ascending.  N is the number of conditioning vectors of a window (29 at
// w = 5000, ~210 at w = 65536), so the scan is a few hundred broadcast loads per thread.
__global__ void k_lc_scatter_det(const aew_lc_scatter_t p) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)p.B * p.N * p.C;
    if (e >= total) return;
    const int c = (int)(e % p.C);
    const int j = (int)((e / p.C) % p.N);
    const int bq = (int)(e / ((int64_t)p.C * p.N));
    float acc = 0.f;
    if (!p.take_compat) {
        const int64_t* jit = p.jitter + (int64_t)bq * p.jit_pitch;
        const float* d = p.d + (int64_t)bq * p.d_bs + c;
        for (int t = 0; t < p.N; ++t) {
            int64_t jj = jit[t];
            jj = jj < 0 ? 0 : (jj > p.N - 1 ? p.N - 1 : jj);
            if (jj == j) acc += d[(int64_t)t * p.d_pitch];
        }
    } else {
        // torch.take flattening (SURVEY C-1): flat = b * N + jitter lands on (flat / (C N), flat % N, (flat / N) % C) =
        // (b / C, jitter, b % C) for every channel of row t, so element (bq, j, c) collects batch element b = bq C + c
        const int64_t b = (int64_t)bq * p.C + c;
        if (b < p.B) {
            const int64_t* jit = p.jitter + b * p.jit_pitch;
            for (int t = 0; t < p.N; ++t) {
                int64_t jj = jit[t];
                jj = jj < 0 ? 0 : (jj > p.N - 1 ? p.N - 1 : jj);
                if (jj != j) continue;
                const float* d = p.d + b * p.d_bs + (int64_t)t * p.d_pitch;
                for (int cc = 0; cc < p.C; ++cc) acc += d[cc];
            }
        }
    }
    p.dsrc[(int64_t)bq * p.dsrc_bs + (int64_t)j * p.dsrc_pitch + c] = acc;
}

__global__ void k_lc_scatter(const aew_lc_scatter_t p) {
    // dsrc must be zeroed by the caller; atomics because several t may hit the same source row
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)p.B * p.N * p.C;
    if (e >= total) return;
    const int c = (int)(e % p.C);
    const int t = (int)((e / p.C) % p.N);
    const int b = (int)(e / ((int64_t)p.C * p.N));
    const float g = p.d[(int64_t)b * p.d_bs + (int64_t)t * p.d_pitch + c];
    int64_t j = p.jitter[(int64_t)b * p.jit_pitch + t];
    j = j < 0 ? 0 : (j > p.N - 1 ? p.N - 1 : j);          // same clamp as k_lc_gather
    if (p.take_compat) {
        const int64_t flat = (int64_t)b * p.N + j;
        const int64_t bb = flat / ((int64_t)p.C * p.N), cc = (flat / p.N) % p.C, nn = flat % p.N;
        atomicAdd(p.dsrc + bb * p.dsrc_bs + nn * p.dsrc_pitch + cc, g);
    } else {
        atomicAdd(p.dsrc + (int64_t)b * p.dsrc_bs + j * p.dsrc_pitch + c, g);
    }
}

// =============================================================================================
// speaker-conditioned gated bias  (wavenet.py:135-139 folded into a per-(batch,layer) bias)
// =============================================================================================
__global__ void k_spk_bias(const aew_spk_bias_t p) {
    // grid: (ceil(2*D_pad/256), L, B)
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int l = blockIdx.y, b = blockIdx.z;
    if (n >= 2 * p.D_pad) return;
    const int gate = (n >> 4) & 1;
    const int co = (n >> 5) * 16 + (n & 15);
    float v = 0.f;
    if (co < p.D) {
        const int64_t ob = gate ? p.off_bias_gate[l] : p.off_bias_sig[l];
        if (ob >= 0) v = p.params[ob + co];
        const int64_t ov = (gate ? p.off_proj_gate[l] : p.off_proj_sig[l]) + (int64_t)co * (p.C_lc + p.G) + p.C_lc;
        const int64_t vb = p.voice[b];
        for (int j = 0; j < p.G; ++j) {
            float gc = p.params[p.off_spk_w + (int64_t)j * p.n_speakers + vb];
            if (p.off_spk_b >= 0) gc += p.params[p.off_spk_b + j];
            v += p.params[ov + j] * gc;
            if (l == 0 && n == 0) p.gc[b * p.G + j] = gc;
        }
    }
    p.bias[((int64_t)b * p.L + l) * 2 * p.D_pad + n] = v;
}

#define AEW_SPK_MAXB 16
#define AEW_SPK_MAXG 16
__global__ void k_spk_bwd(const aew_spk_bwd_t p, int det) {
    // grid (L, 2, ceil(B / 16)): one block per (layer, filt|gate, chunk of 16 batch elements); thread = output channel co.
    //   phase 1: everything a thread needs from global memory is fetched up front (column sums of dfg per batch
    //            element, its row of the speaker projection); it writes its bias / projection gradients and leaves
    //            both vectors in LDS;
    //   phase 2: thread (b, j) sums cs[b][co] * V[co][j] over the channels in ascending order.
    // (Until round 3 phase 2 was B x G cross-lane reductions per wave with LDS atomics - 80 dependent ds_bpermute
    // chains, ~45 us for 40 blocks - and its order depended on the atomics.)
    // B <= 16 is one chunk: plain stores, deterministic.  Larger batches: the chunks add their bias / projection
    // partial sums with atomics into the (pre-zeroed) gradient buffer.
    const int l = blockIdx.x + (p.layer_range & 0xffff), half = blockIdx.y;     // (layer_range = 0: first layer 0)
    const int b0 = blockIdx.z * AEW_SPK_MAXB, nb = min(AEW_SPK_MAXB, p.B - b0);
    const bool multi = gridDim.z > 1;
    const int tid = threadIdx.x;
    extern __shared__ float sh[];                    // gc [nb][G] | cs [nb][257] | vs [G][257]
    float* gcs = sh;
    float* cs = sh + AEW_SPK_MAXB * p.G;
    float* vs = cs + AEW_SPK_MAXB * 257;
    for (int i = tid; i < nb * p.G; i += blockDim.x) gcs[i] = p.gc[b0 * p.G + i];
    const int64_t ob = half ? p.off_bias_gate[l] : p.off_bias_sig[l];
    const int64_t ov0 = half ? p.off_proj_gate[l] : p.off_proj_sig[l];
    const int pb = p.G > 0 ? tid / p.G : nb, pj = tid - pb * p.G;   // phase 2: this thread's (batch element, embedding column)
    float dgc = 0.f;
    __syncthreads();
    for (int co0 = 0; co0 < p.D; co0 += 256) {
        const int co = co0 + tid;
        const bool ok = co < p.D;
        const int n = (co >> 4) * 32 + (co & 15) + 16 * half;
        const int64_t ov = ov0 + (int64_t)co * (p.C_lc + p.G) + p.C_lc;
        // per-batch column sums of dfg (layers below colsum_running: the buffer holds the sums over batch elements 0..b)
        float csv[AEW_SPK_MAXB], vrow[AEW_SPK_MAXG];
#pragma unroll
        for (int b = 0; b < AEW_SPK_MAXB; ++b)
            csv[b] = (ok && b < nb) ? p.colsum[((int64_t)(b0 + b) * p.L + l) * 2 * p.D_pad + n] : 0.f;
#pragma unroll
        for (int j = 0; j < AEW_SPK_MAXG; ++j) vrow[j] = (ok && j < p.G) ? p.params[ov + j] : 0.f;
        const bool running = l < p.colsum_running;
        float bsum = 0.f;
        float prev = (running && ok && b0 > 0) ? p.colsum[((int64_t)(b0 - 1) * p.L + l) * 2 * p.D_pad + n] : 0.f;
#pragma unroll
        for (int b = 0; b < AEW_SPK_MAXB; ++b) {
            const float raw = csv[b];
            if (running && b0 + b > 0 && b < nb) csv[b] -= prev;
            prev = raw;
            bsum += csv[b];
            if (b < nb) cs[b * 257 + tid] = csv[b];
        }
        if (ok && ob >= 0) {
            if (multi) atomicAdd(p.grads + ob + co, bsum);
            else p.grads[ob + co] = bsum;
        }
#pragma unroll
        for (int j = 0; j < AEW_SPK_MAXG; ++j) {
            if (j < p.G) {
                vs[j * 257 + tid] = vrow[j];
                float gv = 0.f;
#pragma unroll
                for (int b = 0; b < AEW_SPK_MAXB; ++b)
                    if (b < nb) gv += csv[b] * gcs[b * p.G + j];
                if (ok) {
                    if (multi) atomicAdd(p.grads + ov + j, gv);
                    else p.grads[ov + j] = gv;
                }
            }
        }
        __syncthreads();
        if (pb < nb) {
            const float* c = cs + pb * 257;
            const float* v = vs + pj * 257;
            const int nco = min(256, p.D - co0);
            for (int k = 0; k < nco; ++k) dgc += c[k] * v[k];
        }
        __syncthreads();
    }
    // speaker embedding grads accumulate over layers (caller zeroes them)
    if (!det) {                                                   // fp32 atomics: the order of the 2 L terms is the schedule's
        if (pb < nb) {
            atomicAdd(p.grads + p.off_spk_w + (int64_t)pj * p.n_speakers + p.voice[b0 + pb], dgc);
            if (p.off_spk_b >= 0) atomicAdd(p.grads + p.off_spk_b + pj, dgc);
        }
        return;
    }
    // Deterministic form: every block leaves its [batch element][embedding column] terms write-through, takes a ticket,
    // and the block that draws the last one adds them - layer ascending, filt before gate, and batch elements ascending
    // where they share a speaker row - as the only writer of the speaker-embedding gradients.
    const int nl = gridDim.x, nz = gridDim.z;
    float* mine = p.det_scratch + (((int64_t)blockIdx.x * 2 + half) * nz + blockIdx.z) * (AEW_SPK_MAXB * AEW_SPK_MAXG);
    if (pb < nb) __hip_atomic_store(mine + pb * AEW_SPK_MAXG + pj, dgc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ unsigned ticket;
    if (tid == 0) ticket = __hip_atomic_fetch_add(p.det_tickets, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (ticket != (unsigned)(nl * 2 * nz) - 1u) return;
    if (tid == 0) __hip_atomic_store(p.det_tickets, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    // thread (batch element bb, column j): its 2 L terms in order (loads 8 at a time), result to LDS; then thread j walks
    // the batch elements in order and adds into the speaker rows (batch elements that share a speaker: ascending)
    float* tot = sh;                                              // [B][G] (B <= 16 per chunk; nz chunks handled in turn)
    float bsum = 0.f;
    for (int z = 0; z < nz; ++z) {
        const int nbz = min(AEW_SPK_MAXB, p.B - z * AEW_SPK_MAXB);
        __syncthreads();
        if (tid < nbz * p.G) {
            const int rb = tid / p.G, j = tid - rb * p.G;
            const float* base = p.det_scratch + (int64_t)z * (AEW_SPK_MAXB * AEW_SPK_MAXG) + rb * AEW_SPK_MAXG + j;
            const int64_t stride = (int64_t)nz * (AEW_SPK_MAXB * AEW_SPK_MAXG);     // term (li, h) sits (li * 2 + h) strides further
            float t = 0.f;
            for (int q = 0; q < 2 * nl; q += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    v[u] = q + u < 2 * nl ? __hip_atomic_load(base + (int64_t)(q + u) * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
#pragma unroll
                for (int u = 0; u < 8; ++u) t += v[u];
            }
            tot[rb * p.G + j] = t;
        }
        __syncthreads();
        if (tid < p.G) {
            // speaker rows: a batch element adds into the running value of the FIRST batch element of its chunk with the same
            // speaker (LDS), and that one writes the row - no dependent global read-modify-write chain
            for (int rb = 0; rb < nbz; ++rb) {
                const int64_t v = p.voice[z * AEW_SPK_MAXB + rb];
                int first = rb;
                for (int r2 = 0; r2 < rb; ++r2)
                    if (p.voice[z * AEW_SPK_MAXB + r2] == v) { first = r2; break; }
                bsum += tot[rb * p.G + tid];
                if (first != rb) tot[first * p.G + tid] += tot[rb * p.G + tid];
            }
            for (int rb = 0; rb < nbz; ++rb) {
                const int64_t v = p.voice[z * AEW_SPK_MAXB + rb];
                bool is_first = true;
                for (int r2 = 0; r2 < rb; ++r2) is_first = is_first && p.voice[z * AEW_SPK_MAXB + r2] != v;
                if (is_first) p.grads[p.off_spk_w + (int64_t)tid * p.n_speakers + v] += tot[rb * p.G + tid];
            }
        }
    }
    if (tid < p.G && p.off_spk_b >= 0) p.grads[p.off_spk_b + tid] += bsum;
}

// =============================================================================================
// base layer: one-hot x 1x1 conv == column gather   (wavenet.py:348-351)
// =============================================================================================
__global__ void k_base_gather(const aew_base_gather_t p) {
    // grid: (ceil(R_pad/4/64), T, B); thread handles 4 channels
    const int c4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int t = blockIdx.y, b = blockIdx.z;
    const int q = (int)p.wav[(int64_t)b * p.wav_pitch + p.wav_off + t];
    if (c4 < p.R_pad) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = c4 + r;
            v[r] = c < p.R ? p.W[(int64_t)c * p.Q + q] + (p.bias ? p.bias[c] : 0.f)
                           : ((p.ones_channel && c == p.R) ? 1.0f : 0.f);
        }
        *reinterpret_cast<uint2*>(p.x + (int64_t)b * p.x_bs + (int64_t)t * p.x_pitch + c4) = pack4_bf16(v);
    }
    if (p.onehot && c4 < p.Q_pad) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (c4 + r == q) ? 1.f : 0.f;
        *reinterpret_cast<uint2*>(p.onehot + (int64_t)b * p.oh_bs + (int64_t)t * p.oh_pitch + c4) = pack4_bf16(v);
    }
}

// fast form: one wave per row, 8 channels per lane, the row of the transposed table Wt[q][:] is one
// contiguous read (the [R][Q] layout costs R loads 1 KiB apart per row: 107 us -> ~25 us at B=8, T=7046)
// A wave takes AEW_BG_ROWS rows: the dependent pair (class of the row -> its table row) is issued for all of them before
// the first store, so a wave has four rows' loads in flight instead of one (54 -> ~30 us for 56 k rows at B = 8).
#define AEW_BG_ROWS 4
__global__ __launch_bounds__(256) void k_base_gather_t(const aew_base_gather_t p) {
    const int lane = threadIdx.x & 63;
    const int t0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * AEW_BG_ROWS, b = blockIdx.y;
    if (t0 >= p.T) return;
    const int c8 = lane * 8;
    int q[AEW_BG_ROWS];
#pragma unroll
    for (int i = 0; i < AEW_BG_ROWS; ++i)
        q[i] = (int)p.wav[(int64_t)b * p.wav_pitch + p.wav_off + min(t0 + i, p.T - 1)];
    float bias8[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) bias8[r] = (p.bias && c8 + r < p.R) ? p.bias[c8 + r] : 0.f;
    if (c8 < p.R_pad) {
        float4 w0[AEW_BG_ROWS], w1[AEW_BG_ROWS];
#pragma unroll
        for (int i = 0; i < AEW_BG_ROWS; ++i) {
            const float* wr = p.Wt + (int64_t)q[i] * p.R_pad + c8;
            w0[i] = *reinterpret_cast<const float4*>(wr);
            w1[i] = *reinterpret_cast<const float4*>(wr + 4);
        }
#pragma unroll
        for (int i = 0; i < AEW_BG_ROWS; ++i) {
            if (t0 + i >= p.T) break;
            float v[8] = {w0[i].x, w0[i].y, w0[i].z, w0[i].w, w1[i].x, w1[i].y, w1[i].z, w1[i].w};
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int c = c8 + r;
                v[r] = c < p.R ? v[r] + bias8[r] : ((p.ones_channel && c == p.R) ? 1.0f : 0.f);
            }
            *reinterpret_cast<uint4*>(p.x + (int64_t)b * p.x_bs + (int64_t)(t0 + i) * p.x_pitch + c8) =
                make_uint4(pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]), pack2_bf16(v[4], v[5]), pack2_bf16(v[6], v[7]));
        }
    }
    if (p.onehot && c8 < p.Q_pad) {
#pragma unroll
        for (int i = 0; i < AEW_BG_ROWS; ++i) {
            if (t0 + i >= p.T) break;
            uint32_t w[4];
#pragma unroll
            for (int r = 0; r < 4; ++r)                               // bf16 1.0 = 0x3f80
                w[r] = (c8 + 2 * r == q[i] ? 0x3f80u : 0u) | (c8 + 2 * r + 1 == q[i] ? 0x3f800000u : 0u);
            *reinterpret_cast<uint4*>(p.onehot + (int64_t)b * p.oh_bs + (int64_t)(t0 + i) * p.oh_pitch + c8) =
                make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
}

// =============================================================================================
// fused log-softmax + NLL and its gradient; one wave per position   (wavenet.py:543-547)
// =============================================================================================
__global__ __launch_bounds__(256) void k_softmax_nll(const aew_softmax_nll_t p) {
    const int lane = threadIdx.x & 63;
    const int64_t pos = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pos >= (int64_t)p.B * p.w) return;
    const int b = (int)(pos / p.w), u = (int)(pos % p.w);
    const float* lg = p.logits + (int64_t)b * p.bs + (int64_t)u * p.pitch;
    const bool live = u < p.w - 1;
    if (p.Q == 256 && p.Q_pad == 256 && (p.pitch & 3) == 0 && (p.bs & 3) == 0 && (!p.backward || ((p.dl_pitch | p.dl_bs) & 3) == 0)) {
        // the reference's 256 classes: the row is ONE 16-byte load per lane, every pass runs on registers and the
        // gradient leaves as one 8-byte store per lane (the general path below re-reads the row per pass, 4 bytes a lane)
        const float4 v4 = *reinterpret_cast<const float4*>(lg + 4 * lane);
        const float v[4] = {v4.x, v4.y, v4.z, v4.w};
        const float mx4 = wave_max(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
        float e[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) e[r] = __expf(v[r] - mx4);
        const float se4 = wave_sum((e[0] + e[1]) + (e[2] + e[3]));
        const float lse4 = mx4 + __logf(se4);
        const int tgt4 = live ? (int)p.wav[(int64_t)b * p.wav_pitch + p.tgt_off + u + 1] : 0;
        if (!p.backward) {
            // the lane that holds the target class reports (its value never leaves the registers)
            if (lane == (tgt4 >> 2)) {
                const int r = tgt4 & 3;
                const float lp = (r == 0 ? v[0] : r == 1 ? v[1] : r == 2 ? v[2] : v[3]) - lse4;
                p.nll[pos] = live ? -lp : 0.f;
                if (p.ptgt) p.ptgt[pos] = live ? __expf(lp) : 0.f;
            }
            if (p.peak) {
                // peak log-probability = max - logsumexp; its class = the lowest one holding the maximum: the first
                // lane (classes ascend with the lane) that has it reports
                const float m01 = fmaxf(v[0], v[1]), m23 = fmaxf(v[2], v[3]);
                const bool has = fmaxf(m01, m23) == mx4;
                const unsigned long long mask = __ballot(has);
                if (lane == __builtin_ctzll(mask)) {
                    const int r = v[0] == mx4 ? 0 : (v[1] == mx4 ? 1 : (v[2] == mx4 ? 2 : 3));
                    p.peak[pos] = -__logf(se4);
                    p.amax[pos] = 4 * lane + r;
                }
            }
        } else {
            uint16_t* dl4 = p.dlogits + (int64_t)b * p.dl_bs + (int64_t)u * p.dl_pitch;
            const float scale4 = p.gmul ? p.scale * p.gmul[0] : p.scale;
            const float inv = 1.0f / se4;                          // exp(v - lse) = e / se
            uint16_t o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float g = live ? (e[r] * inv - ((4 * lane + r) == tgt4 ? 1.f : 0.f)) * scale4 : 0.f;
                o[r] = f2bf(g);
            }
            *reinterpret_cast<uint2*>(dl4 + 4 * lane) = make_uint2((uint32_t)o[0] | ((uint32_t)o[1] << 16), (uint32_t)o[2] | ((uint32_t)o[3] << 16));
        }
        return;
    }
    float mx = -INFINITY;
    for (int c = lane; c < p.Q; c += 64) mx = fmaxf(mx, lg[c]);
    mx = wave_max(mx);
    float se = 0.f;
    for (int c = lane; c < p.Q; c += 64) se += __expf(lg[c] - mx);
    se = wave_sum(se);
    const float lse = mx + __logf(se);
    const int tgt = live ? (int)p.wav[(int64_t)b * p.wav_pitch + p.tgt_off + u + 1] : 0;
    if (!p.backward) {
        if (lane == 0) {
            const float lp = lg[tgt] - lse;
            p.nll[pos] = live ? -lp : 0.f;
            if (p.ptgt) p.ptgt[pos] = live ? __expf(lp) : 0.f;
        }
    } else {
        uint16_t* dl = p.dlogits + (int64_t)b * p.dl_bs + (int64_t)u * p.dl_pitch;
        const float scale = p.gmul ? p.scale * p.gmul[0] : p.scale;
        for (int c = lane; c < p.Q_pad; c += 64) {
            float g = 0.f;
            if (live && c < p.Q) g = (__expf(lg[c] - lse) - (c == tgt ? 1.f : 0.f)) * scale;
            dl[c] = f2bf(g);
        }
    }
}

// =============================================================================================
// column sums (bias gradients).  grid: (ceil(N/256), batch, row chunks); 4 waves stride the rows,
// each lane owns 4 consecutive columns (8/16-byte loads); LDS combine, then one atomic per column.
// =============================================================================================
// W = columns per lane: 8 (bf16, 16-byte loads) or 4 (fp32).  A block covers 64*W columns and
// `rows_per_chunk` rows; its 4 waves stride the rows with 4 independent loads in flight each.
template <int W>
__global__ __launch_bounds__(256) void k_colsum(const aew_colsum_t p, int rows_per_chunk, int det) {
    __shared__ float sh[4][64 * W];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // narrow matrices (N <= 32 W): `lpr` lanes cover a row and a wave takes G = 64 / lpr rows per load (128 bf16
    // columns left 48 of 64 lanes idle: the upsampler bias gradients ran at 0.5 TB/s)
    int lpr = 64;
    if (gridDim.x == 1) { while (lpr > 1 && (lpr >> 1) * W >= p.N) lpr >>= 1; }
    const int G = 64 / lpr, cl = lane & (lpr - 1), rg = lane / lpr;
    const int col = blockIdx.x * (64 * W) + cl * W;
    const int b = blockIdx.y;
    const int r0 = blockIdx.z * rows_per_chunk, r1 = min(p.M, r0 + rows_per_chunk);
    float s[W];
#pragma unroll
    for (int r = 0; r < W; ++r) s[r] = 0.f;
    auto load = [&](int m, float v[W]) {
        const int64_t row = (int64_t)m * p.x.row_step + p.x.row_off;
        const bool ok = m < r1 && row >= p.x.row_lo && row < p.x.row_hi;
        const int64_t idx = (int64_t)b * p.x.batch_stride + (ok ? row : (int64_t)p.x.row_lo) * p.x.row_pitch + col;
        if (W == 8) {
            const uint4 t = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(p.x.ptr) + idx);
            const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                v[2 * q] = ok ? __uint_as_float(w[q] << 16) : 0.f;
                v[2 * q + 1] = ok ? __uint_as_float(w[q] & 0xffff0000u) : 0.f;
            }
        } else {
            const float4 t = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.x.ptr) + idx);
            v[0] = ok ? t.x : 0.f; v[1] = ok ? t.y : 0.f; v[W - 2] = ok ? t.z : 0.f; v[W - 1] = ok ? t.w : 0.f;
        }
    };
    if (col < p.N && p.x.row_hi > p.x.row_lo)
        for (int m = r0 + wv * G + rg; m < r1; m += 16 * G) {
            float v[4][W];
#pragma unroll
            for (int u = 0; u < 4; ++u) load(m + 4 * G * u, v[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int r = 0; r < W; ++r) s[r] += v[u][r];
        }
    for (int o = lpr; o < 64; o <<= 1) {                       // add the G row groups of the wave
#pragma unroll
        for (int r = 0; r < W; ++r) s[r] += __shfl_xor(s[r], o);
    }
#pragma unroll
    for (int r = 0; r < W; ++r) sh[wv][lane * W + r] = (lane < lpr) ? s[r] : 0.f;
    __syncthreads();
    if (!det) {
        for (int c = threadIdx.x; c < 64 * W; c += 256) {
            const int cc = blockIdx.x * (64 * W) + c;
            if (cc < p.N) atomicAdd(p.out + (int64_t)b * p.out_bs + cc, sh[0][c] + sh[1][c] + sh[2][c] + sh[3][c]);
        }
        return;
    }
    // Deterministic form: the partial sums of this (batch element, row chunk) leave write-through, the block takes a
    // ticket of its output - one output per column block when the batch elements share it (out_bs == 0), one per
    // (column block, batch element) otherwise - and the last arriver adds all partials in a FIXED order: the list of
    // partials (batch element ascending, chunk ascending) is cut into NG contiguous ranges, one per thread group, each
    // summed in order (loads 8 at a time), and the NG range sums are added in range order.  One association whatever the
    // schedule, and a single writer per output.
    const int nchunk = gridDim.z, nb = gridDim.y, ncb = gridDim.x;
    const int col0 = blockIdx.x * (64 * W);
    const int ncol = min(64 * W, p.N - col0);                         // valid columns of this column block
    float* mine = p.det_scratch + (((int64_t)b * nchunk + blockIdx.z) * ncb + blockIdx.x) * (64 * W);
    for (int c = threadIdx.x; c < ncol; c += 256)
        __hip_atomic_store(mine + c, sh[0][c] + sh[1][c] + sh[2][c] + sh[3][c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const bool shared_out = p.out_bs == 0;
    unsigned* tk = p.det_tickets + (shared_out ? blockIdx.x : blockIdx.x * nb + b);
    const unsigned want = (unsigned)(shared_out ? nb * nchunk : nchunk) - 1u;
    __shared__ unsigned ticket;
    if (threadIdx.x == 0) ticket = __hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (ticket != want) return;
    if (threadIdx.x == 0) __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
    const int b_lo = shared_out ? 0 : b;
    const int npart = shared_out ? nb * nchunk : nchunk;              // partial q: batch element b_lo + q / nchunk, chunk q % nchunk
    int cpad = 1;
    while (cpad < ncol) cpad <<= 1;                                   // columns rounded up to a power of two <= 64 W <= 512
    const int NG = cpad >= 256 ? 1 : 256 / cpad;                       // thread groups
    const int per = (npart + NG - 1) / NG;
    float* red = &sh[0][0];                                           // [NG][cpad] <= 256 floats ... 4 * 64 W available
    for (int c0 = 0; c0 < cpad; c0 += 256 / NG) {                      // (one pass unless cpad > 256)
        const int c = c0 + (int)(threadIdx.x % (256 / NG)), gq = threadIdx.x / (256 / NG);
        float tot = 0.f;
        if (c < ncol) {
            const int q0 = gq * per, q1 = min(npart, q0 + per);
            const float* base = p.det_scratch + ((int64_t)b_lo * nchunk * ncb + blockIdx.x) * (64 * W) + c;
            const int64_t stride = (int64_t)ncb * (64 * W);           // partial q sits q strides further (b-major, chunk-minor)
            for (int q = q0; q < q1; q += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    v[u] = q + u < q1 ? __hip_atomic_load(base + (int64_t)(q + u) * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
#pragma unroll
                for (int u = 0; u < 8; ++u) tot += v[u];
            }
        }
        __syncthreads();
        red[gq * (256 / NG) + (threadIdx.x % (256 / NG))] = tot;
        __syncthreads();
        if (gq == 0 && c < ncol) {
            float t2 = 0.f;
            for (int k = 0; k < NG; ++k) t2 += red[k * (256 / NG) + (threadIdx.x % (256 / NG))];
            p.out[(int64_t)b * p.out_bs + col0 + c] += t2;
        }
    }
}

__global__ void k_zero(uint4* p, int64_t n16) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) p[i] = make_uint4(0, 0, 0, 0);
}

// =============================================================================================
// scalar reduction (loss):  out[0] = sum_i scale_i * sum(x_i)     single block, deterministic
// =============================================================================================
__global__ __launch_bounds__(1024) void k_reduce(const aew_reduce_t p) {
    // one block of 1024 threads, 4 loads in flight per thread; fixed summation order (deterministic)
    __shared__ float sh[1024];
    float tot = 0.f;
    for (int i = 0; i < p.n_terms; ++i) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        const float* x = p.x[i];
        const int n = p.n[i];
        int e = threadIdx.x;
        for (; e + 3072 < n; e += 4096) { s0 += x[e]; s1 += x[e + 1024]; s2 += x[e + 2048]; s3 += x[e + 3072]; }
        for (; e < n; e += 1024) s0 += x[e];
        sh[threadIdx.x] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        for (int o = 512; o > 0; o >>= 1) {
            if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
            __syncthreads();
        }
        const float v = sh[0] * p.scale[i];
        if (threadIdx.x == 0) p.out[1 + i] = v;
        const float ps = p.post_scale_dev[i] ? p.post_scale_dev[i][0] : p.post_scale[i];
        tot += p.clamp[i] ? ps * fmaxf(v, p.clamp_min[i]) : v;
        __syncthreads();
    }
    if (threadIdx.x == 0) p.out[0] = tot;
}

// =============================================================================================
// mean / unbiased std of a channels-last view (gradient statistics of run(), autoencoder_model.py:252-257)
// =============================================================================================
__global__ __launch_bounds__(1024) void k_moments(const aew_moments_t p) {
    // one block; thread e takes elements e, e+1024, ... of the (batch, row, col) index space; fp64 partial sums
    // combined by a fixed tree -> deterministic and free of the cancellation of sum(x^2) - n mean^2 in fp32
    __shared__ double sh[2][1024];
    const int64_t per_b = (int64_t)p.rows * p.cols, n = per_b * p.batch;
    double s = 0.0, q = 0.0;
    // four elements per thread and pass, loads first (thread t still adds elements t, t + 1024, ... in that order);
    // 32-bit index arithmetic where it fits (two 64-bit divisions per element were most of what this block executed)
    const bool small = n < ((int64_t)1 << 31);
    for (int64_t e0 = threadIdx.x; e0 < n; e0 += 4096) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t e = e0 + 1024 * u;
            v[u] = 0.f;
            if (e >= n) continue;
            int b, r, c;
            if (small) {
                const unsigned e32 = (unsigned)e, pb = (unsigned)per_b, cols = (unsigned)p.cols;
                const unsigned bq = e32 / pb, rem = e32 - bq * pb, rq = rem / cols;
                b = (int)bq; r = (int)rq; c = (int)(rem - rq * cols);
            } else {
                b = (int)(e / per_b);
                r = (int)((e - b * per_b) / p.cols); c = (int)(e - b * per_b - (int64_t)r * p.cols);
            }
            const int row = r * p.x.row_step + p.x.row_off;
            if (row < p.x.row_lo || row >= p.x.row_hi) continue;                // rows outside the view read as zero
            const int64_t off = b * p.x.batch_stride + (int64_t)row * p.x.row_pitch + c;
            v[u] = p.x.dtype == AEW_BF16 ? bf2f(static_cast<const uint16_t*>(p.x.ptr)[off])
                                         : static_cast<const float*>(p.x.ptr)[off];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { s += (double)v[u]; q += (double)v[u] * (double)v[u]; }
    }
    sh[0][threadIdx.x] = s;
    sh[1][threadIdx.x] = q;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            sh[0][threadIdx.x] += sh[0][threadIdx.x + o];
            sh[1][threadIdx.x] += sh[1][threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double S = sh[0][0], Q = sh[1][0], mean = n > 0 ? S / (double)n : 0.0;
        const double var = n > 1 ? fmax(Q - S * mean, 0.0) / (double)(n - 1) : 0.0;
        p.out[0] = (float)mean;
        p.out[1] = (float)sqrt(var);
        p.out[2] = (float)S;
        p.out[3] = (float)Q;
    }
}

// =============================================================================================
// Adam (torch.optim.Adam defaults; checkpoint.py:49-50), flat buffer, float4 vectorised
// =============================================================================================
__global__ void k_adam(const aew_adam_t a) {
    const int64_t i4 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i4 >= a.n) return;
    if (a.guard && *a.guard) return;                   // a chained launch of this step gave up a wait: parameters stay
    const float inv_sqrt_bc2 = rsqrtf(a.bc2);
    const float step = a.lr / a.bc1;
    if (i4 + 4 <= a.n) {
        float4 p = *reinterpret_cast<float4*>(a.p + i4);
        float4 g = *reinterpret_cast<const float4*>(a.g + i4);
        float4 m = *reinterpret_cast<float4*>(a.m + i4);
        float4 v = *reinterpret_cast<float4*>(a.v + i4);
        float* pp = &p.x; float* gp = &g.x; float* mp = &m.x; float* vp = &v.x;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float gr = gp[r] * a.grad_scale;
            mp[r] = a.beta1 * mp[r] + (1.f - a.beta1) * gr;
            vp[r] = a.beta2 * vp[r] + (1.f - a.beta2) * gr * gr;
            pp[r] -= step * mp[r] / (sqrtf(vp[r]) * inv_sqrt_bc2 + a.eps);
        }
        *reinterpret_cast<float4*>(a.p + i4) = p;
        *reinterpret_cast<float4*>(a.m + i4) = m;
        *reinterpret_cast<float4*>(a.v + i4) = v;
    } else {
        for (int64_t i = i4; i < a.n; ++i) {
            const float gr = a.g[i] * a.grad_scale;
            const float m = a.beta1 * a.m[i] + (1.f - a.beta1) * gr;
            const float v = a.beta2 * a.v[i] + (1.f - a.beta2) * gr * gr;
            a.m[i] = m; a.v[i] = v;
            a.p[i] -= step * m / (sqrtf(v) * inv_sqrt_bc2 + a.eps);
        }
    }
}

// =============================================================================================
// VAE reparameterisation + KL terms (vae_bn.py:44-53, 90-98) and AE norm term (ae_bn.py:36-38)
// =============================================================================================
__global__ void k_vae(const aew_vae_t p) {
    const int q = blockIdx.x;
    const int lane = threadIdx.x;                     // 64 threads
    const float* lin = p.lin + (int64_t)q * p.lin_pitch;
    float kl = 0.f;
    const float kl_coef = (p.kl_coef_dev ? p.kl_coef_dev[0] : p.kl_coef) * ((p.backward && p.gmul) ? p.gmul[0] : 1.0f);
    float klc = kl_coef;
    if (p.backward && p.kl_value) klc = (p.kl_value[0] >= p.free_nats) ? kl_coef : 0.f;
    for (int j = lane; j < p.d_pitch; j += 64) {
        if (j < p.d) {
            const float mu = lin[j], ls = lin[p.d + j];
            const float sigma = __expf(0.5f * ls);
            const float e = p.eps[(int64_t)q * p.d + j];
            if (!p.backward) {
                p.sample[(int64_t)q * p.d_pitch + j] = mu + sigma * e;
                const float s2 = sigma * sigma;
                kl += 1.0f + __logf(s2) - mu * mu - s2;
            } else {
                // loss = ... + kl_coef * KL,  KL = -0.5*sum(1 + ls - mu^2 - exp(ls))
                const float ds = p.dsample[(int64_t)q * p.d_pitch + j];
                p.dlin[(int64_t)q * p.lin_pitch + j] = ds + klc * mu;
                p.dlin[(int64_t)q * p.lin_pitch + p.d + j] =
                    ds * e * 0.5f * sigma + klc * (-0.5f) * (1.0f - sigma * sigma);
            }
        } else if (!p.backward) {
            p.sample[(int64_t)q * p.d_pitch + j] = 0.f;
        }
    }
    if (!p.backward) {
        kl = wave_sum(kl);
        if (lane == 0) p.kl_terms[q] = kl;
    }
}

__global__ void k_ae_norm(const aew_ae_norm_t p) {
    const int q = blockIdx.x, lane = threadIdx.x;     // 64 threads
    const float* z = p.ze + (int64_t)q * p.d_pitch;
    float ss = 0.f;
    for (int j = lane; j < p.d; j += 64) ss += z[j] * z[j];
    ss = wave_sum(ss);
    const float nrm = sqrtf(ss);
    if (!p.backward) {
        if (lane == 0) p.term[q] = fabsf(nrm - 1.0f);
    } else {
        const float sgn = nrm > 1.0f ? 1.f : (nrm < 1.0f ? -1.f : 0.f);
        const float coef = p.gmul ? p.coef * p.gmul[0] : p.coef;
        for (int j = lane; j < p.d_pitch; j += 64) {
            float g = 0.f;
            if (j < p.d) g = p.dze_in[(int64_t)q * p.d_pitch + j] + (nrm > 0.f ? coef * sgn * z[j] / nrm : 0.f);
            p.dze[(int64_t)q * p.d_pitch + j] = g;
        }
    }
}

// =============================================================================================
// time-jitter indices (jitter.py:13-33): one thread per batch row (the documented rule is a
// second-order chain; n is a few hundred at most)
// =============================================================================================
__device__ __forceinline__ unsigned long long aew_mix64(unsigned long long z) {      // splitmix64 finaliser
    z ^= z >> 30; z *= 0xbf58476d1ce4e5b9ull;
    z ^= z >> 27; z *= 0x94d049bb133111ebull;
    z ^= z >> 31;
    return z;
}
__device__ __forceinline__ double aew_jitter_u(unsigned long long seed, unsigned long long step, int b, int t) {
    unsigned long long h = aew_mix64(seed + 0x9e3779b97f4a7c15ull);
    h = aew_mix64(h ^ (step + 0x9e3779b97f4a7c15ull));
    h = aew_mix64(h ^ (((unsigned long long)(unsigned)b << 32) | (unsigned long long)(unsigned)t));
    return (double)(h >> 11) * (1.0 / 9007199254740992.0);                               // 53 bits -> [0, 1)
}
__global__ void k_jitter(const aew_jitter_t j) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= j.B) return;
    const double p = (double)j.p, s = 1.0 - 2.0 * (double)j.p;
    int x2 = 1, x1 = 1;
    for (int t = 0; t < j.n; ++t) {
        int x = 1;
        if (t >= 2) {
            const double u = aew_jitter_u(j.seed, j.step, b, t);
            double c0 = p, c1 = p + s;                                   // cumulative [p, s, p]
            if (j.mode == 1 && x2 == 2 && x1 == 1) { c0 = 0.0; c1 = s / (p + s); }
            x = (u >= c0 ? 1 : 0) + (u >= c1 ? 1 : 0);
        }
        j.out[(int64_t)b * j.out_pitch + t] = (int64_t)(t - 1 + x);
        x2 = x1; x1 = x;
    }
}

// =============================================================================================
// per-step diagnostics (vqema_bn.py:155-160, 251-264; util.py:98-105) as two small kernels
// =============================================================================================
// scratch: double[2] = sum, sum of squares of the peak log-probability; float[256] at byte 16 = arg-max counts
__global__ __launch_bounds__(256) void k_diag_peak(const aew_vq_diag_t p) {
    __shared__ float bins[256];
    __shared__ double part[4][2];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    bins[threadIdx.x] = 0.f;
    __syncthreads();
    const int64_t n_pos = (int64_t)p.B * (p.w - 1);
    double s1 = 0.0, s2 = 0.0;
    // 16 lanes per position, 4 positions per wave and iteration, the row read ONCE into registers (the first version
    // walked one position per wave with two dependent passes over its row: 159 us for 40 k positions)
    const int sub = lane >> 4, l16 = lane & 15;
    const int per = (p.n_quant + 15) >> 4;                        // <= 16 classes per lane, contiguous
    for (int64_t pos0 = ((int64_t)blockIdx.x * 4 + wv) * 4; pos0 < n_pos; pos0 += (int64_t)gridDim.x * 16) {
        const int64_t pos = pos0 + sub;
        const bool live = pos < n_pos;
        const int64_t pc = live ? pos : n_pos - 1;
        const int b = (int)(pc / (p.w - 1)), u = (int)(pc % (p.w - 1));
        const float* lg = p.logits + (int64_t)b * p.bs + (int64_t)u * p.pitch + l16 * per;
        float v[16];
        float mx = -INFINITY;
        int am = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            v[k] = (k < per && l16 * per + k < p.n_quant) ? lg[k] : -INFINITY;
            if (v[k] > mx) { mx = v[k]; am = l16 * per + k; }       // first maximum of the lane (ascending class)
        }
        float gmx = mx;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) gmx = fmaxf(gmx, __shfl_xor(gmx, o, 16));
        int cand = (mx == gmx) ? am : 0x7fffffff;                 // lowest class among the lanes holding the maximum
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) cand = min(cand, __shfl_xor(cand, o, 16));
        float se = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) se += __expf(v[k] - gmx);     // exp(-inf) = 0 for the padding
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) se += __shfl_xor(se, o, 16);
        const float pk = -__logf(se);                             // max - logsumexp
        if (live && l16 == 0) {
            s1 += (double)pk; s2 += (double)pk * (double)pk;
            atomicAdd(&bins[cand & 255], 1.f);
        }
    }
#pragma unroll
    for (int o = 16; o < 64; o <<= 1) {                           // the four position slots of the wave
        s1 += __shfl_xor(s1, o);
        s2 += __shfl_xor(s2, o);
    }
    if (lane == 0) { part[wv][0] = s1; part[wv][1] = s2; }
    __syncthreads();
    double* acc = reinterpret_cast<double*>(p.scratch);
    float* gb = reinterpret_cast<float*>(reinterpret_cast<char*>(p.scratch) + 16);
    if (threadIdx.x == 0) {
        atomicAdd(acc, part[0][0] + part[1][0] + part[2][0] + part[3][0]);
        atomicAdd(acc + 1, part[0][1] + part[1][1] + part[2][1] + part[3][1]);
    }
    if (bins[threadIdx.x] != 0.f) atomicAdd(gb + threadIdx.x, bins[threadIdx.x]);
}

__device__ __forceinline__ float block_red(float v, bool is_max, float* sh) {    // 1024 threads
    v = is_max ? wave_max(v) : -wave_max(-v);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wv] = v;
    __syncthreads();
    float r = sh[0];
    for (int i = 1; i < 16; ++i) r = is_max ? fmaxf(r, sh[i]) : fminf(r, sh[i]);
    return r;
}
__device__ __forceinline__ double block_sum(double v, double* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wv] = v;
    __syncthreads();
    double r = 0.0;
    for (int i = 0; i < 16; ++i) r += sh[i];
    return r;
}
// The bulk of the diagnostics in AEW_DIAG_PARTS blocks: a single block reads the 1 MB codebook (or 2 x 160 KB of
// per-position values) at the fill rate of one CU (~23 us); the slices leave 16 floats each in `scratch`
//   [0..3] min / max row norm of the ze slice, of the codebook slice   [4..7] two doubles: sum / sum of squares of the peak
//   [8..15] 256-bit set of the arg-max classes seen
// and k_diag_final combines them in slice order (deterministic).
#define AEW_DIAG_PARTS 16
__global__ __launch_bounds__(256) void k_diag_part(const aew_vq_diag_t p) {
    __shared__ float shf[4][2];
    __shared__ double shd[4][2];
    __shared__ int seen[256];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, b = blockIdx.x;
    float* part = reinterpret_cast<float*>(p.scratch) + b * 16;
    auto norms = [&](const float* base, int rows, int pitch, int slot) {
        const int per = (rows + AEW_DIAG_PARTS - 1) / AEW_DIAG_PARTS;
        const int r_lo = b * per, r_hi = min(rows, r_lo + per);
        float lo = INFINITY, hi = -INFINITY;
        const int sub = tid & 7;
        for (int r0 = r_lo; r0 < r_hi; r0 += 256) {           // 8 lanes per row, 8 rows per thread in flight
            float ss[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = r0 + u * 32 + (tid >> 3);
                const float* x = base + (int64_t)min(r, rows - 1) * pitch;
                float a = 0.f;
                if ((p.d & 3) == 0 && (pitch & 3) == 0) {
                    for (int j = sub * 4; j < p.d; j += 32) {
                        const float4 v = *reinterpret_cast<const float4*>(x + j);
                        a += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
                    }
                } else {
                    for (int j = sub; j < p.d; j += 8) a += x[j] * x[j];
                }
                ss[u] = a;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = r0 + u * 32 + (tid >> 3);
                float a = ss[u];
                a += __shfl_xor(a, 1); a += __shfl_xor(a, 2); a += __shfl_xor(a, 4);
                if (r < r_hi) { const float nr = sqrtf(a); lo = fminf(lo, nr); hi = fmaxf(hi, nr); }
            }
        }
        lo = -wave_max(-lo); hi = wave_max(hi);
        __syncthreads();
        if (lane == 0) { shf[wv][0] = lo; shf[wv][1] = hi; }
        __syncthreads();
        if (tid == 0) {
            part[slot] = fminf(fminf(shf[0][0], shf[1][0]), fminf(shf[2][0], shf[3][0]));
            part[slot + 1] = fmaxf(fmaxf(shf[0][1], shf[1][1]), fmaxf(shf[2][1], shf[3][1]));
        }
    };
    if (p.ze) norms(p.ze, p.Q, p.d_pitch, 0);
    if (p.emb) norms(p.emb, p.K, p.d, 2);
    if (p.peak && p.amax) {
        seen[tid] = 0;
        __syncthreads();
        const int64_t n_all = (int64_t)p.B * p.w;
        const int64_t per = (n_all + AEW_DIAG_PARTS - 1) / AEW_DIAG_PARTS;
        const int64_t i_lo = b * per, i_hi = min(n_all, i_lo + per);
        double s1 = 0.0, s2 = 0.0;
        for (int64_t i0 = i_lo + tid; i0 < i_hi; i0 += 1024) {
            float pk[4];
            int am[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t i = min(i0 + 256 * u, n_all - 1);
                pk[u] = p.peak[i]; am[u] = p.amax[i];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t i = i0 + 256 * u;
                if (i >= i_hi || (int)(i % p.w) == p.w - 1) continue;
                const double v = (double)pk[u];
                s1 += v; s2 += v * v;
                seen[am[u] & 255] = 1;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
        __syncthreads();
        if (lane == 0) { shd[wv][0] = s1; shd[wv][1] = s2; }
        __syncthreads();
        const unsigned long long bits = __ballot(seen[tid] != 0);       // wave wv: classes 64 wv .. 64 wv + 63
        if (lane == 0) {
            reinterpret_cast<uint32_t*>(part)[8 + 2 * wv] = (uint32_t)bits;
            reinterpret_cast<uint32_t*>(part)[9 + 2 * wv] = (uint32_t)(bits >> 32);
        }
        if (tid == 0) {
            double* dp = reinterpret_cast<double*>(part + 4);
            dp[0] = (shd[0][0] + shd[1][0]) + (shd[2][0] + shd[3][0]);
            dp[1] = (shd[0][1] + shd[1][1]) + (shd[2][1] + shd[3][1]);
        }
    }
}

__global__ __launch_bounds__(1024) void k_diag_final(const aew_vq_diag_t p, int parts) {
    __shared__ float shf[16];
    __shared__ double shd[16];
    const int tid = threadIdx.x;
    float o[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // row norms: 8 lanes per row, one float4 each per 32 channels (a thread-per-row walk touches 64 cache lines per
    // load instruction and made this single block the whole cost of the op: 0.13 ms for 4096 codes)
    auto norms = [&](const float* base, int rows, int pitch, float& lo, float& hi) {
        lo = INFINITY; hi = -INFINITY;
        const int part = tid & 7;
        if (p.d == 64 && (pitch & 3) == 0) {
            // the codebook's shape: NR rows per thread and pass, their 2 NR loads in flight together (one row per pass
            // left this single block waiting for a memory round trip 32 times for 4096 codes)
            constexpr int NR = 8;
            for (int r0 = 0; r0 < rows; r0 += 128 * NR) {
                float4 v[NR][2];
#pragma unroll
                for (int u = 0; u < NR; ++u) {
                    const int r = r0 + u * 128 + (tid >> 3);
                    const float* x = base + (int64_t)min(r, rows - 1) * pitch + part * 4;
                    v[u][0] = *reinterpret_cast<const float4*>(x);
                    v[u][1] = *reinterpret_cast<const float4*>(x + 32);
                }
#pragma unroll
                for (int u = 0; u < NR; ++u) {
                    const int r = r0 + u * 128 + (tid >> 3);
                    float ss = v[u][0].x * v[u][0].x + v[u][0].y * v[u][0].y + v[u][0].z * v[u][0].z + v[u][0].w * v[u][0].w;
                    ss += v[u][1].x * v[u][1].x + v[u][1].y * v[u][1].y + v[u][1].z * v[u][1].z + v[u][1].w * v[u][1].w;
                    ss += __shfl_xor(ss, 1); ss += __shfl_xor(ss, 2); ss += __shfl_xor(ss, 4);
                    if (r < rows) { const float nr = sqrtf(ss); lo = fminf(lo, nr); hi = fmaxf(hi, nr); }
                }
            }
            return;
        }
        for (int r0 = 0; r0 < rows; r0 += 128) {
            const int r = r0 + (tid >> 3);
            float ss = 0.f;
            if (r < rows) {
                const float* x = base + (int64_t)r * pitch;
                if ((p.d & 3) == 0 && (pitch & 3) == 0) {
                    for (int j = part * 4; j < p.d; j += 32) {
                        const float4 v = *reinterpret_cast<const float4*>(x + j);
                        ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
                    }
                } else {
                    for (int j = part; j < p.d; j += 8) ss += x[j] * x[j];
                }
            }
            ss += __shfl_xor(ss, 1); ss += __shfl_xor(ss, 2); ss += __shfl_xor(ss, 4);
            if (r < rows) { const float nr = sqrtf(ss); lo = fminf(lo, nr); hi = fmaxf(hi, nr); }
        }
    };
    // code counts: in registers before anything else (<= 4 per thread), so that their memory round trips run under the
    // norm passes instead of after them
    const bool pre = p.K <= 4096;
    float hv[4] = {0.f, 0.f, 0.f, 0.f}, nv[4] = {0.f, 0.f, 0.f, 0.f};
    if (pre) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = tid + 1024 * u;
            if (p.hist && k < p.K) hv[u] = p.hist[k];
            if (p.n_sum && k < p.K) nv[u] = p.n_sum[k];
        }
    }
    const float* part = reinterpret_cast<const float*>(p.scratch);       // parts > 0: the slices' results (k_diag_part)
    if (p.ze) {
        float lo = INFINITY, hi = -INFINITY;
        if (parts) { if (tid < parts) { lo = part[tid * 16]; hi = part[tid * 16 + 1]; } }
        else norms(p.ze, p.Q, p.d_pitch, lo, hi);
        o[0] = block_red(lo, false, shf); o[1] = block_red(hi, true, shf);
    }
    if (p.emb) {
        float lo = INFINITY, hi = -INFINITY;
        if (parts) { if (tid < parts) { lo = part[tid * 16 + 2]; hi = part[tid * 16 + 3]; } }
        else norms(p.emb, p.K, p.d, lo, hi);
        o[2] = block_red(lo, false, shf); o[3] = block_red(hi, true, shf);
    }
    if (p.hist) {                                             // -sum n log2 n, n = hist / sum(hist); 0 log 0 = 0
        double s = 0.0;
        if (pre) { for (int u = 0; u < 4; ++u) s += (double)hv[u]; }          // k = tid, tid + 1024, ...: the same order
        else for (int k = tid; k < p.K; k += 1024) s += (double)p.hist[k];
        const double tot = block_sum(s, shd);
        double e = 0.0;
        if (pre) {
            for (int u = 0; u < 4; ++u) {
                const float n = (float)((double)hv[u] / tot);
                if (n > 0.f) e -= (double)(n * log2f(n));
            }
        } else {
            for (int k = tid; k < p.K; k += 1024) {
                const float n = (float)((double)p.hist[k] / tot);
                if (n > 0.f) e -= (double)(n * log2f(n));
            }
        }
        o[4] = (float)block_sum(e, shd);
    }
    if (p.n_sum) {
        double c = 0.0;
        if (pre) { for (int u = 0; u < 4; ++u) c += nv[u] > 0.f ? 1.0 : 0.0; }
        else for (int k = tid; k < p.K; k += 1024) c += p.n_sum[k] > 0.f ? 1.0 : 0.0;
        o[5] = (float)block_sum(c, shd);
    }
    if (p.peak && p.amax) {                                   // per-position arrays written by the softmax kernel
        __shared__ int seen[256];
        if (tid < 256) seen[tid] = 0;
        __syncthreads();
        double s1 = 0.0, s2 = 0.0;
        const int64_t n_all = (int64_t)p.B * p.w;
        if (parts) {                                          // slice sums in slice order (thread 0), class sets OR-ed
            if (tid == 0)
                for (int q = 0; q < parts; ++q) {
                    const double* dp = reinterpret_cast<const double*>(part + q * 16 + 4);
                    s1 += dp[0]; s2 += dp[1];
                }
            if (tid < 256) {
                uint32_t w = 0;
                for (int q = 0; q < parts; ++q) w |= reinterpret_cast<const uint32_t*>(part)[q * 16 + 8 + (tid >> 5)];
                seen[tid] = (w >> (tid & 31)) & 1;
            }
        } else if (n_all < (int64_t)1 << 31) {
            // four positions per thread and pass (loads first), 32-bit index arithmetic; thread t still adds positions
            // t, t + 1024, ... in that order
            const unsigned n32 = (unsigned)n_all, w32 = (unsigned)p.w;
            for (unsigned i0 = tid; i0 < n32; i0 += 4096) {
                float pk[4];
                int am[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const unsigned i = min(i0 + 1024u * u, n32 - 1);
                    pk[u] = p.peak[i]; am[u] = p.amax[i];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const unsigned i = i0 + 1024u * u;
                    if (i >= n32 || i % w32 == w32 - 1) continue;
                    const double v = (double)pk[u];
                    s1 += v; s2 += v * v;
                    seen[am[u] & 255] = 1;
                }
            }
        } else {
            for (int64_t i = tid; i < n_all; i += 1024) {
                if ((int)(i % p.w) == p.w - 1) continue;
                const double pk = (double)p.peak[i];
                s1 += pk; s2 += pk * pk;
                seen[p.amax[i] & 255] = 1;
            }
        }
        const double t1 = block_sum(s1, shd), t2 = block_sum(s2, shd);
        const double n = (double)p.B * (p.w - 1);
        const double mean = t1 / n;
        o[6] = (float)mean;
        o[7] = n > 1.0 ? (float)sqrt(fmax((t2 - n * mean * mean) / (n - 1.0), 0.0)) : 0.f;   // torch.std: unbiased
        __syncthreads();
        o[8] = (float)block_sum(tid < 256 && seen[tid] ? 1.0 : 0.0, shd);
    } else if (p.logits) {
        const double* acc = reinterpret_cast<const double*>(p.scratch);
        const float* gb = reinterpret_cast<const float*>(reinterpret_cast<const char*>(p.scratch) + 16);
        const double n = (double)p.B * (p.w - 1);
        const double mean = acc[0] / n;
        o[6] = (float)mean;
        o[7] = n > 1.0 ? (float)sqrt(fmax((acc[1] - n * mean * mean) / (n - 1.0), 0.0)) : 0.f;   // torch.std: unbiased
        double c = tid < 256 && gb[tid] > 0.f ? 1.0 : 0.0;
        o[8] = (float)block_sum(c, shd);
    }
    if (tid < 9) p.out[tid] = o[tid];
}

// =============================================================================================
// MFCC front-end (mfcc.py:39-76).  Tiny and latency-bound: a B x 74-frame batch is 93 MFLOP.
// =============================================================================================
// grid (n_frames, B): one frame per block.  scratch: [B][n_frames][n_mels] dB | [B][n_frames] frame max
__global__ __launch_bounds__(256) void k_mfcc_mel(const aew_mfcc_t p) {
    __shared__ float s[1024];
    __shared__ float pw[520];
    __shared__ float red[4];
    const int f = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int Ly = p.left_pad + p.n;
    const float* wav = p.wav + (int64_t)b * p.wav_bs;
    for (int n = tid; n < p.win; n += 256) {
        int i = f * p.hop + n - p.win / 2;                    // index into y, reflected at both ends (center=True)
        if (i < 0) i = -i;
        if (i >= Ly) i = 2 * (Ly - 1) - i;
        const float v = (i >= p.left_pad && i < Ly && i >= 0) ? wav[i - p.left_pad] : 0.f;
        s[n] = v * p.window[n];
    }
    __syncthreads();
    for (int k = tid; k < p.n_bins; k += 256) {
        float re = 0.f, im = 0.f;
        int idx = 0;
        for (int n = 0; n < p.win; ++n) {
            const float2 w = reinterpret_cast<const float2*>(p.twiddle)[idx];
            re += s[n] * w.x;
            im -= s[n] * w.y;
            idx += k;
            if (idx >= p.win) idx -= p.win;
        }
        pw[k] = re * re + im * im;
    }
    __syncthreads();
    float db = -INFINITY;
    if (tid < p.n_mels) {
        const float* w = p.melw + (int64_t)tid * p.n_bins;
        float a = 0.f;
        for (int k = 0; k < p.n_bins; ++k) a += w[k] * pw[k];
        db = 10.f * log10f(fmaxf(a, 1e-10f));
        p.scratch[((int64_t)b * p.n_frames + f) * p.n_mels + tid] = db;
    }
    const float m = wave_max(db);
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    if (tid == 0)
        p.scratch[(int64_t)p.B * p.n_frames * p.n_mels + (int64_t)b * p.n_frames + f] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// grid (B): top_db floor, DCT, trim, derivatives
__global__ __launch_bounds__(256) void k_mfcc_finish(const aew_mfcc_t p) {
    __shared__ float red[4];
    __shared__ float thr_s;
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* db = p.scratch + (int64_t)b * p.n_frames * p.n_mels;
    const float* fmx = p.scratch + (int64_t)p.B * p.n_frames * p.n_mels + (int64_t)b * p.n_frames;
    float* cep = p.scratch + (int64_t)p.B * p.n_frames * (p.n_mels + 1) + (int64_t)b * p.n_frames * p.n_mfcc;   // [n_mfcc][n_frames]
    float m = -INFINITY;
    for (int f = tid; f < p.n_frames; f += 256) m = fmaxf(m, fmx[f]);
    m = wave_max(m);
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    if (tid == 0) thr_s = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) - 80.f;
    __syncthreads();
    const float thr = thr_s;
    for (int e = tid; e < p.n_frames * 16; e += 256) {
        const int f = e >> 4, c = e & 15;
        if (c >= p.n_mfcc) continue;
        const float* d = p.dct + (int64_t)c * p.n_mels;
        const float* x = db + (int64_t)f * p.n_mels;
        float a = 0.f;
        for (int j = 0; j < p.n_mels; ++j) a += d[j] * fmaxf(x[j], thr);
        cep[(int64_t)c * p.n_frames + f] = a;
    }
    __syncthreads();                                          // (block-scope visibility of the global writes above)
    const int Ft = p.n_frames - p.trim_left - p.trim_right;
    float* out = p.out + (int64_t)b * p.out_bs;
    for (int e = tid; e < Ft * 16; e += 256) {
        const int f = e >> 4, c = e & 15;
        if (c >= p.n_mfcc) continue;
        const float* x = cep + (int64_t)c * p.n_frames + p.trim_left;       // trimmed series of coefficient c
        out[(int64_t)c * p.out_pitch + f] = x[f];
#pragma unroll
        for (int o = 0; o < 2; ++o) {                         // Savitzky-Golay width 9, mode='interp' at the edges
            const float* sg = p.sg + o * 81;
            float a = 0.f;
            if (f < 4) {
                for (int j = 0; j < 9; ++j) a += sg[9 + f * 9 + j] * x[j];
            } else if (f >= Ft - 4) {
                for (int j = 0; j < 9; ++j) a += sg[45 + (f - (Ft - 4)) * 9 + j] * x[Ft - 9 + j];
            } else {
                for (int j = 0; j < 9; ++j) a += sg[j] * x[f - 4 + j];
            }
            out[(int64_t)((o + 1) * p.n_mfcc + c) * p.out_pitch + f] = a;
        }
    }
}

// =============================================================================================
// launchers
// =============================================================================================
static inline unsigned cdiv64(int64_t a, int64_t b) { return (unsigned)((a + b - 1) / b); }

static int launch_copy(const aew_copy_table_t& t, hipStream_t st) {
    if (t.n_blocks <= 0) return 0;
    hipLaunchKernelGGL(k_copy_table, dim3(t.n_blocks), dim3(256), 0, st, t);
    return (int)hipGetLastError();
}
static int launch_vq_nearest(const aew_vq_nearest_t& p, hipStream_t st) {
    if (p.scratch && p.n_split > 1) {
        if (p.n_split > 64 || ((uintptr_t)p.scratch & 7)) return AEW_E_ARG;
        const int per = (p.K + p.n_split - 1) / p.n_split;
        const bool al = (((uintptr_t)p.emb) & 15) == 0;
        if (per <= 256 && al && p.d == 32 && p.Q >= 16)
            hipLaunchKernelGGL((k_vq_nearest_part_mq<32, 8>), dim3(p.n_split, (p.Q + 7) / 8), dim3(256), 0, st, p);
        else if (per <= 256 && al && p.d == 64 && p.Q >= 16)
            hipLaunchKernelGGL((k_vq_nearest_part_mq<64, 8>), dim3(p.n_split, (p.Q + 7) / 8), dim3(256), 0, st, p);
        else
            hipLaunchKernelGGL(k_vq_nearest_part, dim3(p.n_split, p.Q), dim3(256), 0, st, p);
        hipLaunchKernelGGL(k_vq_nearest_combine, dim3(p.Q), dim3(64), 0, st, p);
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL(k_vq_nearest, dim3(p.Q), dim3(256), 0, st, p);
    return (int)hipGetLastError();
}
static int launch_zero(const aew_zero_t& z, hipStream_t st);
static int launch_vq_stats(const aew_vq_stats_t& p, hipStream_t st) {
    if (p.Q > 0 && p.Q <= 1024 && p.d <= 64 && (int64_t)p.K * p.d >= 16 * (int64_t)p.Q) {
        // few queries, many codes: clear the accumulators, then one wave per query (k_vq_stats_few)
        aew_zero_t z1 = {p.z_sum, (int64_t)p.K * p.d * 4}, z2 = {p.n_sum, (int64_t)p.K * 4};
        int rc;
        if (p.n_sum == p.z_sum + (int64_t)p.K * p.d) {         // the engine allocates them back to back: one launch
            z1.bytes += z2.bytes;
            rc = launch_zero(z1, st);
        } else {
            rc = launch_zero(z1, st);
            if (rc == 0) rc = launch_zero(z2, st);
        }
        if (rc) return rc;
        hipLaunchKernelGGL(k_vq_stats_few, dim3((p.Q + 3) / 4), dim3(256), 0, st, p);
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL(k_vq_stats, dim3(cdiv64((int64_t)p.K * p.d, 256)), dim3(256), 0, st, p);
    return (int)hipGetLastError();
}
static int launch_vq_ema(const aew_vq_ema_t& p, hipStream_t st) {
    hipLaunchKernelGGL(k_vq_ema, dim3(cdiv64((int64_t)p.K * p.d, 256)), dim3(256), 0, st, p);
    hipLaunchKernelGGL(k_vq_ema_denom, dim3(cdiv64(p.K, 256)), dim3(256), 0, st, p);
    return (int)hipGetLastError();
}
static int launch_vq_bwd(const aew_vq_bwd_t& p, hipStream_t st) {
    hipLaunchKernelGGL(k_vq_bwd, dim3(cdiv64(p.Q, 4)), dim3(256), 0, st, p, AEW_T().deterministic ? 1 : 0);
    return (int)hipGetLastError();
}
static int launch_lc_gather(const aew_lc_gather_t& p, hipStream_t st) {
    hipLaunchKernelGGL(k_lc_gather, dim3(cdiv64((int64_t)p.B * p.N * p.C_pad, 256)), dim3(256), 0, st, p);
    return (int)hipGetLastError();
}
#define AEW_LC_SCATTER_DET_MAXN 4096
static int launch_lc_scatter(const aew_lc_scatter_t& p, hipStream_t st) {
    if (p.N <= AEW_LC_SCATTER_DET_MAXN) {                      // every element of dsrc[b][j][0:C] is written: no zeroing needed
        hipLaunchKernelGGL(k_lc_scatter_det, dim3(cdiv64((int64_t)p.B * p.N * p.C, 256)), dim3(256), 0, st, p);
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL(k_lc_scatter, dim3(cdiv64((int64_t)p.B * p.N * p.C, 256)), dim3(256), 0, st, p);
    return (int)hipGetLastError();
}
static int launch_spk_bias(const aew_spk_bias_t& p, hipStream_t st) {
    hipLaunchKernelGGL(k_spk_bias, dim3(cdiv64(2 * p.D_pad, 256), p.L, p.B), dim3(256), 0, st, p);
    return (int)hipGetLastError();
}
static int launch_spk_bwd(const aew_spk_bwd_t& p, hipStream_t st) {
    if (p.G > AEW_SPK_MAXG || p.B < 1) return AEW_E_UNSUP;     // (engine.DecoderPlan refuses such a model at build time)
    const int l0 = p.layer_range & 0xffff, ln = p.layer_range ? (p.layer_range >> 16) & 0x7fff : p.L;
    if (ln < 1 || l0 + ln > p.L) return AEW_E_ARG;
    const int det = AEW_T().deterministic && p.det_scratch && p.det_tickets;
    hipLaunchKernelGGL(k_spk_bwd, dim3(ln, 2, (p.B + AEW_SPK_MAXB - 1) / AEW_SPK_MAXB), dim3(256),
                       (AEW_SPK_MAXB * p.G + (AEW_SPK_MAXB + p.G) * 257) * sizeof(float), st, p, det);
    return (int)hipGetLastError();
}
static int launch_base_gather(const aew_base_gather_t& p, hipStream_t st) {
    const int cmax = p.onehot && p.Q_pad > p.R_pad ? p.Q_pad : p.R_pad;
    if (p.Wt && cmax <= 512 && p.R_pad % 8 == 0 && (!p.onehot || p.Q_pad % 8 == 0) && (p.x_pitch % 8) == 0 &&
        (!p.onehot || p.oh_pitch % 8 == 0)) {
        hipLaunchKernelGGL(k_base_gather_t, dim3(cdiv64(p.T, 4 * AEW_BG_ROWS), p.B), dim3(256), 0, st, p);
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL(k_base_gather, dim3(cdiv64(cmax / 4, 64), p.T, p.B), dim3(64), 0, st, p);
    return (int)hipGetLastError();
}
static int launch_vq_diag(const aew_vq_diag_t& p, hipStream_t st) {
    if (!p.out || (p.logits && (!p.scratch || p.n_quant < 1 || p.n_quant > 256 || p.w < 2))) return AEW_E_ARG;
    if ((p.peak != nullptr) != (p.amax != nullptr) || (p.peak && p.w < 2)) return AEW_E_ARG;
    if (p.logits && !p.peak) {
        hipError_t e = hipMemsetAsync(p.scratch, 0, 16 + 256 * sizeof(float), st);
        if (e != hipSuccess) return (int)e;
        const int64_t n_pos = (int64_t)p.B * (p.w - 1);
        hipLaunchKernelGGL(k_diag_peak, dim3((unsigned)min((int64_t)512, cdiv64(n_pos, 16))), dim3(256), 0, st, p);
    }
    // codebook norms / per-position statistics in slices first (needs the scratch buffer: 16 floats per slice)
    const int parts = (p.scratch && !(p.logits && !p.peak) && (p.ze || p.emb || p.peak)) ? AEW_DIAG_PARTS : 0;
    if (parts) hipLaunchKernelGGL(k_diag_part, dim3(parts), dim3(256), 0, st, p);
    hipLaunchKernelGGL(k_diag_final, dim3(1), dim3(1024), 0, st, p, parts);
    return (int)hipGetLastError();
}
static int launch_mfcc(const aew_mfcc_t& p, hipStream_t st) {
    if (!p.wav || !p.out || !p.scratch || !p.window || !p.twiddle || !p.melw || !p.dct || !p.sg) return AEW_E_ARG;
    if (p.win < 2 || p.win > 1024 || p.n_bins != p.win / 2 + 1 || p.n_mels < 1 || p.n_mels > 128 || p.n_mfcc < 1 ||
        p.n_mfcc > 16 || p.hop < 1 || p.B < 1)
        return AEW_E_ARG;
    if (p.n_frames != 1 + (p.left_pad + p.n) / p.hop || p.left_pad + p.n <= p.win / 2) return AEW_E_ARG;
    const int Ft = p.n_frames - p.trim_left - p.trim_right;
    if (Ft < 9 || Ft > p.out_pitch) return AEW_E_ARG;
    hipLaunchKernelGGL(k_mfcc_mel, dim3(p.n_frames, p.B), dim3(256), 0, st, p);
    hipLaunchKernelGGL(k_mfcc_finish, dim3(p.B), dim3(256), 0, st, p);
    return (int)hipGetLastError();
}
static int launch_softmax(const aew_softmax_nll_t& p, hipStream_t st) {
    hipLaunchKernelGGL(k_softmax_nll, dim3(cdiv64((int64_t)p.B * p.w, 4)), dim3(256), 0, st, p);
    return (int)hipGetLastError();
}
static void colsum_grid(const aew_colsum_t& p, int& rpc, int& chunks, int& ncb) {
    rpc = 64;                                                // rows per block: 16 per wave, 4 loads in flight
    const int W = p.dtype == AEW_BF16 ? 8 : 4;
    int lpr = 64;                                            // narrow matrices: G rows per wave-load (see k_colsum)
    if (p.N <= 64 * W) { while (lpr > 1 && (lpr >> 1) * W >= p.N) lpr >>= 1; }
    rpc = 64 * (64 / lpr);
    chunks = (p.M + rpc - 1) / rpc;
    ncb = (int)cdiv64(p.N, 64 * W);
}
static int launch_colsum(const aew_colsum_t& p, hipStream_t st) {
    if (!p.accumulate) {
        // contiguous or shared outputs are cleared with one memset; strided per-batch outputs one
        // per batch (callers on the hot path pre-zero the buffer and pass accumulate = 1)
        const int nset = p.out_bs == 0 ? 1 : p.batch;
        for (int b = 0; b < nset; ++b) {
            hipError_t e = hipMemsetAsync(p.out + (int64_t)b * p.out_bs, 0, (size_t)p.N * sizeof(float), st);
            if (e != hipSuccess) return (int)e;
        }
    }
    int rpc, chunks, ncb;
    colsum_grid(p, rpc, chunks, ncb);
    const int det = AEW_T().deterministic && p.det_scratch && p.det_tickets;
    if (p.dtype == AEW_BF16) {
        if ((p.x.row_pitch % 8) || ((uintptr_t)p.x.ptr & 15) || (p.x.batch_stride % 8)) return AEW_E_ALIGN;
        hipLaunchKernelGGL(k_colsum<8>, dim3(ncb, p.batch, chunks), dim3(256), 0, st, p, rpc, det);
    } else {
        hipLaunchKernelGGL(k_colsum<4>, dim3(ncb, p.batch, chunks), dim3(256), 0, st, p, rpc, det);
    }
    return (int)hipGetLastError();
}
extern "C" int aew_colsum_det_size(const aew_colsum_t* c, int64_t* floats, int32_t* tickets) {
    if (!c || !floats || !tickets || c->M < 0 || c->N < 1 || c->batch < 1) return AEW_E_ARG;
    int rpc, chunks, ncb;
    colsum_grid(*c, rpc, chunks, ncb);
    const int W = c->dtype == AEW_BF16 ? 8 : 4;
    *floats = (int64_t)c->batch * (chunks > 0 ? chunks : 1) * ncb * 64 * W;
    *tickets = ncb * c->batch;
    return 0;
}
static int launch_reduce(const aew_reduce_t& p, hipStream_t st) {
    if (p.n_terms < 1 || p.n_terms > 4) return AEW_E_ARG;
    hipLaunchKernelGGL(k_reduce, dim3(1), dim3(1024), 0, st, p);
    return (int)hipGetLastError();
}
static int launch_moments(const aew_moments_t& p, hipStream_t st) {
    if (!p.x.ptr || !p.out || p.rows < 0 || p.cols < 0 || p.batch < 0 || p.cols > p.x.row_pitch) return AEW_E_ARG;
    if (p.x.dtype != AEW_BF16 && p.x.dtype != AEW_F32) return AEW_E_UNSUP;
    hipLaunchKernelGGL(k_moments, dim3(1), dim3(1024), 0, st, p);
    return (int)hipGetLastError();
}
static int launch_adam(const aew_adam_t& p, hipStream_t st) {
    if (((uintptr_t)p.p | (uintptr_t)p.g | (uintptr_t)p.m | (uintptr_t)p.v) & 15) return AEW_E_ALIGN;
    hipLaunchKernelGGL(k_adam, dim3(cdiv64((p.n + 3) / 4, 256)), dim3(256), 0, st, p);
    return (int)hipGetLastError();
}
static int launch_zero(const aew_zero_t& z, hipStream_t st) {
    if (z.bytes <= 0) return 0;
    if (((uintptr_t)z.ptr & 15) || (z.bytes & 15)) return (int)hipMemsetAsync(z.ptr, 0, (size_t)z.bytes, st);
    const int64_t n16 = z.bytes / 16;
    int blocks = (int)((n16 + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_zero, dim3(blocks), dim3(256), 0, st, reinterpret_cast<uint4*>(z.ptr), n16);
    return (int)hipGetLastError();
}
static int launch_vae(const aew_vae_t& p, hipStream_t st) {
    hipLaunchKernelGGL(k_vae, dim3(p.Q), dim3(64), 0, st, p);
    return (int)hipGetLastError();
}
static int launch_jitter(const aew_jitter_t& j, hipStream_t st) {
    if (!j.out || j.B <= 0 || j.n < 0 || j.out_pitch < j.n || !(j.p >= 0.f && j.p <= 0.5f)) return AEW_E_ARG;
    hipLaunchKernelGGL(k_jitter, dim3(cdiv64(j.B, 64)), dim3(64), 0, st, j);
    return (int)hipGetLastError();
}
static int launch_ae_norm(const aew_ae_norm_t& p, hipStream_t st) {
    hipLaunchKernelGGL(k_ae_norm, dim3(p.Q), dim3(64), 0, st, p);
    return (int)hipGetLastError();
}
