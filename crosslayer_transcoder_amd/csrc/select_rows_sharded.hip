// Per-row TopK (k of F_global per (token, layer) row) when the feature axis is sharded over ranks (gfx950).
// Replaces model/topk.py:53-63 for PerLayerTopK under feature sharding; the reference's DTensor plumbing cannot
// shard TopK modules at all (model/parallel.py:268-274).
//
// Every rank first takes its LOCAL top-k of each row (one-block-per-row radix select, compact (index, value) rows of
// the positive entries): the global top-k of a row is a subset of the union of these.  The exact global k-th value of
// every row is then found by an 8-round, 4-bit MSB-first radix select over the compact rows, one all-reduce(sum) of
// the [rows][16] bucket counts per round.  Values tied at the k-th value are taken in GLOBAL feature order, which for
// contiguous feature shards is rank order: an all-gather of the per-row local tie counts tells each rank how many of
// its (index-ordered) ties it keeps.  Finally the kept entries are packed into the rank-local CSR / CSC lists.
#include "select_common.h"
#include "lists_common.h"

namespace {
struct RowState { uint32_t prefix, mask, krem, flags; };      // flags bit0: take every positive entry (fewer than k exist)
struct RHead { uint32_t long_rows, long_cols; uint32_t pad[62]; };

__global__ void rows_init_kernel(RowState* __restrict__ state, int rows, uint32_t k) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < rows) { RowState st; st.prefix = 0; st.mask = 0; st.krem = k; st.flags = 0; state[r] = st; }
}

// counts[row][b] = local compact entries matching the row's prefix whose next 4 key bits equal b (one wave per row)
__global__ __launch_bounds__(TPB) void rows_bucket_kernel(const float* __restrict__ cval, const uint32_t* __restrict__ row_cnt, int k,
                                                          int rows, const RowState* __restrict__ state,
                                                          uint32_t* __restrict__ counts, int round) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63, shift = 28 - 4 * round;
    const RowState st = state[row];
    const uint32_t n = row_cnt[row];
    uint32_t mine = 0;                                         // lane b (< 16) accumulates bucket b
    for (uint32_t base = 0; base < n; base += 64) {
        const uint32_t e = base + lane;
        uint32_t key = 0; bool ok = false;
        if (e < n) { key = f32_key(cval[(long)row * k + e]); ok = (key & st.mask) == st.prefix; }
        const uint32_t b = (key >> shift) & 15u;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const uint32_t c = (uint32_t)__popcll(__ballot(ok && b == (uint32_t)q));
            if (lane == q) mine += c;
        }
    }
    if (lane < 16) counts[(long)row * 16 + lane] = mine;
}

// consume the GLOBAL bucket counts of `round`: narrow the row's prefix to the bucket holding its krem-th largest value
__global__ void rows_find_kernel(RowState* __restrict__ state, const uint32_t* __restrict__ counts, int rows, int round) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows) return;
    RowState st = state[row];
    if (st.flags & 1u) return;
    const int shift = 28 - 4 * round;
    uint32_t c[16], tot = 0;
#pragma unroll
    for (int b = 0; b < 16; ++b) { c[b] = counts[(long)row * 16 + b]; tot += c[b]; }
    if (tot < st.krem) { st.flags |= 1u; state[row] = st; return; }      // only at round 0: fewer than k positives anywhere
    uint32_t cum = 0;
#pragma unroll
    for (int b = 15; b >= 0; --b) {
        if (cum + c[b] >= st.krem) {
            st.prefix |= (uint32_t)b << shift; st.mask |= 15u << shift; st.krem -= cum;
            break;
        }
        cum += c[b];
    }
    state[row] = st;
}

// local entries equal to the row's k-th value T (after the last round prefix == T)
__global__ __launch_bounds__(TPB) void rows_tie_kernel(const float* __restrict__ cval, const uint32_t* __restrict__ row_cnt, int k, int rows,
                                                       const RowState* __restrict__ state, uint32_t* __restrict__ ties_mine) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const RowState st = state[row];
    const uint32_t n = row_cnt[row];
    uint32_t c = 0;
    for (uint32_t e = lane; e < n; e += 64) c += f32_key(cval[(long)row * k + e]) == st.prefix;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if (lane == 0) ties_mine[row] = (st.flags & 1u) ? 0u : c;
}

// keep: value > T, plus this rank's share of the ties (krem ties in global feature order = rank order, then index
// order inside the rank: the compact rows are ascending in feature id).  Rewrites the compact row in place (prefix
// compaction inside one wave) and accumulates the column counts.
__global__ __launch_bounds__(TPB) void rows_apply_kernel(int32_t* __restrict__ cidx, float* __restrict__ cval, uint32_t* __restrict__ row_cnt,
                                                         int k, int rows, int F, int L, const RowState* __restrict__ state,
                                                         const uint32_t* __restrict__ ties_all, int world, int rank,
                                                         uint32_t* __restrict__ col_cnt) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const RowState st = state[row];
    const uint32_t n = row_cnt[row];
    const bool all = st.flags & 1u;
    uint32_t before = 0;
    for (int q = 0; q < rank; ++q) before += ties_all[(long)q * rows + row];
    const uint32_t share = all ? 0u : (st.krem > before ? st.krem - before : 0u);    // ties this rank may keep
    uint32_t kept = 0, ties_seen = 0;
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (uint32_t base = 0; base < n; base += 64) {
        const uint32_t e = base + lane;
        int32_t id = 0; float v = 0.0f; uint32_t key = 0;
        if (e < n) { id = cidx[(long)row * k + e]; v = cval[(long)row * k + e]; key = f32_key(v); }
        const bool tie = e < n && !all && key == st.prefix;
        const unsigned long long tm = __ballot(tie);
        const uint32_t ord = ties_seen + (uint32_t)__popcll(tm & lt);
        const bool keep = e < n && (all || key > st.prefix || (tie && ord < share));
        const unsigned long long km = __ballot(keep);
        __builtin_amdgcn_wave_barrier();
        if (keep) {
            const uint32_t pos = kept + (uint32_t)__popcll(km & lt);      // pos <= e: in-place compaction is safe
            cidx[(long)row * k + pos] = id; cval[(long)row * k + pos] = v;
            atomicAdd(&col_cnt[(row % L) * F + id], 1u);
        }
        kept += (uint32_t)__popcll(km);
        ties_seen += (uint32_t)__popcll(tm);
    }
    if (lane == 0) row_cnt[row] = kept;
}

struct RPlan {
    int rows, ncols, kk;
    size_t off_state, off_counts, off_ties, off_rowcnt, off_colcnt, off_colcur, zero_end, off_long, off_cidx, off_cval, total;
};
RPlan rplan(int B, int L, int F, int k, int world) {
    RPlan p;
    p.rows = B * L; p.ncols = L * F; p.kk = k < F ? k : F;
    size_t o = sizeof(RHead);
    auto take = [&](size_t bytes) { const size_t at = o; o = align_up(o + bytes, 256); return at; };
    p.off_state = take((size_t)p.rows * sizeof(RowState));
    p.off_counts = take((size_t)p.rows * 16 * 4);
    p.off_ties = take((size_t)world * p.rows * 4);
    p.off_rowcnt = take((size_t)p.rows * 4);
    p.off_colcnt = take((size_t)p.ncols * 4);
    p.off_colcur = take((size_t)p.ncols * 4);
    p.zero_end = o;
    p.off_long = take((size_t)(p.rows > p.ncols ? p.rows : p.ncols) * 4);
    p.off_cidx = take((size_t)p.rows * p.kk * 4);
    p.off_cval = take((size_t)p.rows * p.kk * 4);
    p.total = o;
    return p;
}
}  // namespace

extern "C" {
size_t clt_select_rows_sharded_workspace_bytes(int B, int L, int F, int k, int world) {
    if (B <= 0 || L <= 0 || F <= 0 || k < 0 || world < 1) return 0;
    return rplan(B, L, F, k, world).total;
}

size_t clt_select_rows_sharded_exchange_info(int B, int L, int F, int k, int world, int which) {
    if (B <= 0 || L <= 0 || F <= 0 || k < 0 || world < 1) return 0;
    const RPlan p = rplan(B, L, F, k, world);
    switch (which) {
        case 0: return p.off_counts;                       // uint32 [rows*16]: SUM over ranks after stages 0..7
        case 1: return (size_t)p.rows * 16;
        case 2: return p.off_ties;                         // uint32 [world][rows]: ALL-GATHER after stage 8
        case 3: return (size_t)p.rows;
        default: return 0;
    }
}

// stage 0: local top-k rows + bucket counts of round 0;  stages 1..7: find(r-1) + counts(r);  stage 8: find(7) + local
// tie counts;  stage 9: keep / drop + list build.
int clt_select_rows_sharded_f32(int stage_lo, int stage_hi, int world, int rank, int F_global, const float* x,
                                const uint8_t* col_mask, int B, int L, int F, int k, uint32_t cap, uint32_t* row_start,
                                int32_t* ent_col, float* ent_val, uint32_t* col_start, int32_t* csc_tok, float* csc_val,
                                int32_t* csc_ent, uint8_t* fired, uint32_t* total, int32_t* overflow, void* ws, size_t wsb,
                                void* stp) {
    if (B <= 0 || L <= 0 || F <= 0 || k < 0 || cap == 0 || world < 1 || rank < 0 || rank >= world) return CLT_ERR_SHAPE;
    if (stage_lo < 0 || stage_hi > 9 || stage_lo > stage_hi) return CLT_ERR_SHAPE;
    if (k > F_global) return CLT_ERR_K_GT_F;
    if ((unsigned long long)L * F >= 0x7fffffffull || (unsigned long long)B * L >= 0x7fffffffull) return CLT_ERR_RANGE;
    const RPlan p = rplan(B, L, F, k, world);
    if (p.kk > 2048) return CLT_ERR_RANGE;
    if (!ws || wsb < p.total) return CLT_ERR_WORKSPACE;
    hipStream_t st = as_stream(stp);
    char* w8 = (char*)ws;
    RHead* head = (RHead*)w8;
    RowState* state = (RowState*)(w8 + p.off_state);
    uint32_t* counts = (uint32_t*)(w8 + p.off_counts);
    uint32_t* ties = (uint32_t*)(w8 + p.off_ties);
    uint32_t* row_cnt = (uint32_t*)(w8 + p.off_rowcnt);
    uint32_t* col_cnt = (uint32_t*)(w8 + p.off_colcnt);
    uint32_t* col_cur = (uint32_t*)(w8 + p.off_colcur);
    int32_t* long_list = (int32_t*)(w8 + p.off_long);
    int32_t* cidx = (int32_t*)(w8 + p.off_cidx);
    float* cval = (float*)(w8 + p.off_cval);
    const int rows = p.rows, ncols = p.ncols;
    const dim3 wgrid((rows + 3) / 4), tgrid((rows + TPB - 1) / TPB);
    for (int stage = stage_lo; stage <= stage_hi; ++stage) {
        if (stage == 0) {
            if (hipMemsetAsync(overflow, 0, 4, st) != hipSuccess) return CLT_ERR_LAUNCH;
            if (hipMemsetAsync(ws, 0, p.zero_end, st) != hipSuccess) return CLT_ERR_LAUNCH;
            if (k > 0) {
                const int rc = clt_internal_topk_rows(x, nullptr, rows, F, p.kk, cidx, cval, col_mask, L, row_cnt, nullptr, st);
                if (rc != CLT_OK) return rc;
            }
            hipLaunchKernelGGL(rows_init_kernel, tgrid, dim3(TPB), 0, st, state, rows, (uint32_t)k);
            hipLaunchKernelGGL(rows_bucket_kernel, wgrid, dim3(TPB), 0, st, cval, row_cnt, p.kk, rows, state, counts, 0);
        } else if (stage <= 7) {
            hipLaunchKernelGGL(rows_find_kernel, tgrid, dim3(TPB), 0, st, state, counts, rows, stage - 1);
            hipLaunchKernelGGL(rows_bucket_kernel, wgrid, dim3(TPB), 0, st, cval, row_cnt, p.kk, rows, state, counts, stage);
        } else if (stage == 8) {
            hipLaunchKernelGGL(rows_find_kernel, tgrid, dim3(TPB), 0, st, state, counts, rows, 7);
            hipLaunchKernelGGL(rows_tie_kernel, wgrid, dim3(TPB), 0, st, cval, row_cnt, p.kk, rows, state, ties + (size_t)rank * rows);
        } else {
            hipLaunchKernelGGL(rows_apply_kernel, wgrid, dim3(TPB), 0, st, cidx, cval, row_cnt, p.kk, rows, F, L, state, ties, world,
                               rank, col_cnt);
            launch_scan(row_cnt, rows, row_start, total, (uint8_t*)nullptr, (const int32_t*)nullptr, 0, (uint32_t*)long_list, st);   // (long_list is idle until the row sort)
            hipLaunchKernelGGL(compact_rows_kernel, wgrid, dim3(TPB), 0, st, cidx, cval, p.kk, F, L, rows, row_start, cap, ent_col,
                               ent_val, overflow);
            launch_scan(col_cnt, ncols, col_start, nullptr, fired, nullptr, 0, (uint32_t*)long_list, st);   // (long_list idle until sort_cols)
            hipLaunchKernelGGL(fill_cols_kernel, wgrid, dim3(TPB), 0, st, row_start, ent_col, ent_val, rows, L, cap, col_start, col_cur,
                               csc_tok, csc_val, csc_ent, (const int32_t*)nullptr, 0);
            hipLaunchKernelGGL(sort_cols_kernel, dim3((ncols + TPB - 1) / TPB), dim3(TPB), 0, st, col_start, ncols, csc_tok, csc_val,
                               csc_ent, long_list, &head->long_cols, (const int32_t*)nullptr, 0);
            launch_sort_queued_cols(st, col_start, csc_tok, csc_val, csc_ent, long_list, &head->long_cols, (const int32_t*)nullptr, 0);
        }
    }
    CLT_LAUNCH_CHECK(); return CLT_OK;
}
}  // extern "C"
