// jh_tall.hip -- the TALL fast path of JetBlock_df! / JetBlock_df'! (src/Jets.jl:1010-1057) and the fused A'oA (530-534 over (A', A)):
// k_tall_diag_fwd, k_tall_diag_adj (MODE 0 adjoint, MODE 1 fused normal operator; MIXED: rows of any elementwise kind), the split-row fold,
// the normal-equations pass of the device-resident CG loops, their launch rules, the lazy per-operator measurement of the forward's grid walk.
// One of the translation units jh_blockop.hip was split into in round 5 (jh_blockop_common.h).
//
// Data layout in HBM: the range vector d is one slab, block i at element offset row_off[i]; a DIAG block's coefficients are a device array
// of the block's length; the domain vector m of a tall (one-column) operator is a plain array (src/Jets.jl:927).
//  Forward: a workgroup owns an element tile, keeps its m tile in registers and streams `fwd_group` blocks through it (a read once, d written
//  once, m re-read from L2/MALL).  Adjoint: a thread owns 16-byte element vectors and walks the rows IN ORDER, product rounded then added --
//  the reference's `_m .+= mul!(mtmp, op', _d)` (1049) without the mtmp round trip -- bit-identical to the sequential CPU loop.
#include "jh_blockop_common.h"
#include <mutex>
#include <map>

namespace {

// ------------------------------------------------------------------ tall fast path ------------
// 1-D grid of ntiles * ngroups workgroups, walked in BANDS of `band` row groups: inside a band the row
// group is the fastest index (workgroups sharing an m tile are dispatched together, so the tile is
// re-read from L2/MALL, not HBM), bands follow one another.  band = 1 is the fully sequential sweep
// (one block row at a time); band = ngroups touches every row concurrently.  n_scalars % NS == 0.
// MIXED: the rows are not all plain diagonals -- a row may be IDENTITY, SCALE, a (conjugated) DIAG, the Jacobian of a SQUARE
// child, or a ZERO block, which the linear loop SKIPS (src/Jets.jl:1022): its d_i stays as found.  The kind is read from the
// row table (uniform per workgroup: scalar branches), the arithmetic is the general kernels' apply_block_loaded.
template <typename S, int E, int NS, int U, bool NT, int BLK, bool MIXED = false>
__global__ __launch_bounds__(BLK) void k_tall_diag_fwd(const jh_dev_block *__restrict__ blocks, int64_t nrow, int rows_per_wg,
                                                       const S *__restrict__ a_base, int64_t a_stride,
                                                       const S *__restrict__ m, S *__restrict__ d, int64_t n_scalars,
                                                       unsigned ntiles, unsigned ngroups, unsigned band, unsigned ctiles, int fmode = 0)
{
    // fmode (MIXED only; round 5, last session): JetBlock_f! of a tall NONLINEAR operator (src/Jets.jl:1003: d_i = F_i(m), every child overwrites its row -- a
    // zero block writes zeros, nothing is skipped); a SQUARE child squares its input, linear children apply df!
    typedef typename vec_of<S, NS>::type V;
    unsigned tile, grp;
    if (ctiles) {
        // COLUMN bands (late round 4): `ctiles` consecutive tiles of one row group, then the same tiles of the next group, ... then the next
        // band of tiles.  Inside a group's share of a band the workgroups stream linearly like a copy (32-64 tiles = 128-256 KiB), the band
        // of m is reused by every row from L2: at 128-512 rows +3 ... +10 % over the row-concurrent walk, whose consecutive workgroups are a
        // whole block apart (tools/micro/fwd_small_rows.hip, profiles/exp_r04_fwd_small_rows.txt); at 1024 rows the row-concurrent walk wins
        const unsigned per_c = ctiles * ngroups;              // workgroups in a full column band
        const unsigned cb = blockIdx.x / per_c;
        const unsigned r = blockIdx.x - cb * per_c;
        const unsigned cw = (cb * ctiles + ctiles <= ntiles) ? ctiles : ntiles - cb * ctiles;   // last band may be narrower
        grp = r / cw;
        tile = cb * ctiles + r % cw;
    } else {
        const unsigned per_band = ntiles * band;              // workgroups in a full band
        const unsigned b = blockIdx.x / per_band;
        const unsigned r = blockIdx.x - b * per_band;
        const unsigned width = (b * band + band <= ngroups) ? band : ngroups - b * band;   // last band may be narrower
        tile = r / width;
        grp = b * band + r % width;
    }
    const int64_t s0 = ((int64_t)tile * U * BLK + threadIdx.x) * NS;
    const int64_t i0 = (int64_t)grp * rows_per_wg;
    const int64_t i1 = (i0 + rows_per_wg < nrow) ? i0 + rows_per_wg : nrow;
    const bool full = ((int64_t)(tile + 1) * U * BLK * NS) <= n_scalars;
    V mv[U];
    if constexpr (MIXED) {
        // (round 5, session 3) these instantiations also serve rows that are NOT whole, 16-byte aligned packs -- blocks of 101^3 elements in one slab
        // (tall_unaligned_ok): every access goes through the under-aligned helpers, the row's last pack is loaded from n - NS and stored by st_pack
        bool ok[U];
        int64_t sk[U];
#pragma unroll
        for (int k = 0; k < U; k++) {
            ok[k] = (s0 + (int64_t)k * BLK * NS) < n_scalars;
            sk[k] = ok[k] ? pack_start<NS>(s0 + (int64_t)k * BLK * NS, n_scalars) : 0;
            mv[k] = ldu<false, S, NS>(m + sk[k]);
        }
        // late round 5: a tall operator with MANY zero rows (muted shots) walks the list of its non-zero rows (a_stride < 0: a_base is that list and nrow
        // its length) -- a workgroup per zero row cost more than the rows that do something: 1024 x 128^3 with one row in eight 0.83 -> see DESIGN 3.6
        const int *rows = a_stride < 0 ? reinterpret_cast<const int *>(a_base) : nullptr;
        jh_dev_block nxt;                                                                  // the row table one row ahead (scalar loads)
        int64_t inext = (i0 < i1) ? (rows ? (int64_t)rows[i0] : i0) : 0;
        if (i0 < i1) nxt = blocks[inext];
        for (int64_t ii = i0; ii < i1; ii++) {
            const jh_dev_block blk = nxt;
            const int64_t i = inext;
            if (ii + 1 < i1) { inext = rows ? (int64_t)rows[ii + 1] : ii + 1; nxt = blocks[inext]; }
            if (blk.kind == JH_OP_ZERO && !fmode) continue;                                // (1022; f!: JopZeroBlock's d .= 0, 942)
            const bool rc = block_reads_coeff(blk, fmode != 0);
            S *di = d + i * n_scalars;
#pragma unroll
            for (int k = 0; k < U; k++) {
                const V c = rc ? ldu<NT, S, NS>((const S *)blk.coeff + sk[k]) : (V)(S)0;
                // (the store stays a streaming store whatever NT says: temporal stores of a range vector larger than the caches leave dirty lines whose
                // write-back lands on the NEXT kernel -- 256 x 255^3: forward alone 5.82 -> 5.75 ms, forward + adjoint pair 11.8 -> 12.2)
                if (ok[k]) st_pack<true, S, NS>(di, s0 + (int64_t)k * BLK * NS, sk[k], apply_block_loaded<S, E, NS, V>(blk, mv[k], c, false, fmode != 0));   // (1026 / 1003)
            }
        }
        return;
    }
    if (full) {
#pragma unroll
        for (int k = 0; k < U; k++) mv[k] = ld<false>(reinterpret_cast<const V *>(m + s0 + (int64_t)k * BLK * NS));
#pragma unroll 2
        for (int64_t i = i0; i < i1; i++) {
            const S *a = a_base ? a_base + i * a_stride : (const S *)blocks[i].coeff;
            S *di = d + i * n_scalars;
            V av[U];
#pragma unroll
            for (int k = 0; k < U; k++) av[k] = ld<NT>(reinterpret_cast<const V *>(a + s0 + (int64_t)k * BLK * NS));
#pragma unroll
            for (int k = 0; k < U; k++)
                st<NT>(reinterpret_cast<V *>(di + s0 + (int64_t)k * BLK * NS), vmul<S, E, NS, V>(av[k], mv[k], false));
        }
    } else {
        // the last tile of a row: a pack past the end re-reads pack 0 and stores nothing (every mv[k] is defined on every lane:
        // conditionally loaded ones made the compiler keep the tile in scratch, 400 bytes per lane at 8 packs x 1024 threads --
        // tools/kernel_resources.py; tests/test_kernel_resources.py keeps every kernel of the library at 0 bytes of scratch)
        bool ok[U];
        int64_t sk[U];
#pragma unroll
        for (int k = 0; k < U; k++) {
            ok[k] = (s0 + (int64_t)k * BLK * NS) < n_scalars;
            sk[k] = ok[k] ? s0 + (int64_t)k * BLK * NS : 0;
            mv[k] = ld<false>(reinterpret_cast<const V *>(m + sk[k]));
        }
        for (int64_t i = i0; i < i1; i++) {
            const S *a = a_base ? a_base + i * a_stride : (const S *)blocks[i].coeff;
            S *di = d + i * n_scalars;
#pragma unroll
            for (int k = 0; k < U; k++) {
                const V av = ld<NT>(reinterpret_cast<const V *>(a + sk[k]));
                if (ok[k]) st<NT>(reinterpret_cast<V *>(di + sk[k]), vmul<S, E, NS, V>(av, mv[k], false));
            }
        }
    }
}

// ---- the forward of rows OFF the 16-byte grid, lanes anchored on each row's OWN grid (round 6) -------------------------------------------------------
// k_tall_diag_fwd<MIXED> cuts every row into packs by ELEMENT index: pack p = elements [p NS, p NS + NS) of the row.  With an odd block length row i
// starts (i n s) mod 16 bytes into a pack of the slab, so three rows in four are read and written through accesses that straddle two 16-byte slots -- a
// wave's request touches nine 128-byte lines instead of eight, the ninth being the next wave's first (256 x 255^3: forward 0.65-0.68 of the roofline,
// traffic x1.027; a misaligned STORE stream alone costs 5 %: profiles/exp_r05_misalign.txt).  Here a lane owns an ALIGNED pack of the range slab: the
// elements [p NS - ph, p NS - ph + NS) of row i, ph = the row's phase in its 16-byte slot.  Rows i, i + NS, i + 2 NS, ... share their phase, so a
// workgroup takes its rows from ONE phase class (row = class + NS * k) and loads its pack of the model -- under-aligned, from L2 -- once for all of them.
// A diagonal that lives in a slab laid out like the range vector (one allocation for all diagonals) has the same phase: its loads are aligned too; any other
// coefficient array is read under-aligned as before.  A row's first and last slot are partial: those two lanes go element by element.  Same products, one
// rounding each: the bits of k_tall_diag_fwd.
template <typename S, int E, int NS, bool NT, int BLK>
__global__ __launch_bounds__(BLK) void k_tall_fwd_anchored(const jh_dev_block *__restrict__ blocks, int64_t nrow, int rows_per_wg, const S *__restrict__ m,
                                                           S *__restrict__ d, int64_t n_scalars, unsigned ntiles, unsigned ngroups, unsigned ctiles,
                                                           int base_phase, int fmode)
{
    typedef typename vec_of<S, NS>::type V;
    // column bands as in k_tall_diag_fwd: `ctiles` consecutive tiles of one row group, then the same tiles of the next group
    const unsigned per_c = ctiles * ngroups;
    const unsigned cb = blockIdx.x / per_c;
    const unsigned r = blockIdx.x - cb * per_c;
    const unsigned cw = (cb * ctiles + ctiles <= ntiles) ? ctiles : ntiles - cb * ctiles;
    const unsigned grp = r / cw, tile = cb * ctiles + r % cw;
    // group -> (phase class q, chunk of rows_per_wg rows of that class): rows q + NS * (chunk * rows_per_wg + j)
    const unsigned q = grp % NS, chunk = grp / NS;
    const int ph = (int)(((int64_t)base_phase + (int64_t)q * n_scalars) % NS);       // scalars of the row's first slot that belong to the row BEFORE it
    const int64_t p = (int64_t)tile * BLK + threadIdx.x;                               // the lane's slot of the row
    const int64_t e_lo = p * NS - ph;                                                  // first element of the slot (negative in the row's first slot)
    const bool any = e_lo < n_scalars && e_lo + NS > 0;
    const bool whole = e_lo >= 0 && e_lo + NS <= n_scalars;
    const int64_t ec = e_lo < 0 ? 0 : (e_lo + NS <= n_scalars ? e_lo : n_scalars - NS);   // a pack inside the row to load from (partial slots: clamped)
    const V mv = ldu<false, S, NS>(m + ec);
    for (int j = 0; j < rows_per_wg; j++) {
        const int64_t i = (int64_t)q + (int64_t)NS * ((int64_t)chunk * rows_per_wg + j);
        if (i >= nrow) break;
        const jh_dev_block blk = blocks[i];
        if (blk.kind == JH_OP_ZERO && !fmode) continue;                                // (1022; f!: JopZeroBlock's d .= 0, 942)
        const bool rc = block_reads_coeff(blk, fmode != 0);
        S *di = d + i * n_scalars;
        const V c = rc ? ldu<NT, S, NS>((const S *)blk.coeff + ec) : (V)(S)0;
        if (!any) continue;
        const V o = apply_block_loaded<S, E, NS, V>(blk, mv, c, false, fmode != 0);    // (1026 / 1003)
        if (whole) {
            stu<true, S, NS>(di + e_lo, o);                                            // (an aligned address by construction)
        } else {
            // the row's first / last slot: the pack was loaded from `ec`; store the scalars of [e_lo, e_lo + NS) that lie in the row.  Complex elements
            // never straddle a slot boundary's ownership here: the scalars are written one by one, each from its own position of the pack
#pragma unroll
            for (int k = 0; k < NS; k++) {
                const int64_t e = ec + k;
                if (e >= e_lo && e < e_lo + NS && e >= 0 && e < n_scalars) st<true>(di + e, (S)o[k]);
            }
        }
    }
}

// one thread: U vectors of the domain, all rows in order.  MODE 0: adjoint (reads a_i, d_i);
// MODE 1: fused normal equations y = sum_i conj(a_i) .* (a_i .* m) (reads a_i only).
// TAIL (MIXED only): the instantiation also serves rows that are not whole, 16-byte aligned packs (a partial last pack).  False for the ONE shape whose
// registers do not hold the extra state -- 1024 lanes x 4 packs x 4 rows of the fused A'A (128 VGPRs per lane; with the tail logic it spilled 12-28 bytes) --
// which therefore keeps aligned operators only; off the grid that size runs 4 packs x 2 rows (launch_tall_adj_mixed)
template <typename S, int E, int NS, int U, int DEPTH, bool NT, int MODE, int BLK, bool MIXED = false, bool TAIL = MIXED>
__global__ __launch_bounds__(BLK) void k_tall_diag_adj(const jh_dev_block *__restrict__ blocks, int64_t nrow,
                                                       const S *__restrict__ a_base, int64_t a_stride, S *__restrict__ out,
                                                       const S *__restrict__ in, int64_t n_scalars, int direct,
                                                       int64_t s_begin, int64_t s_end, int64_t row0, int64_t row1, int accumulate,
                                                       int64_t rows_per_part, S *__restrict__ part_out, int64_t part_stride, S in_scale = (S)1)
{
    // in_scale (MIXED, MODE 0): every d_i is multiplied by it, rounded, before A_i' is applied -- (a * A)' d = A'(conj(a) d) of the scalar-times-operator chain
    // (src/Jets.jl:1160) in one pass for operators with rows of several kinds; 1 is exact (x * 1 == x bit for bit), so the plain adjoint pays one multiply
    // rows [row0, row1) of the operator; accumulate != 0 continues the ordered sum from what `out` holds (a long operator
    // can be walked in several launches with the bits of one: ((0 + p_0) + p_1) + ... is the same sequence)
    // the launch covers the scalar range [s_begin, s_end) of the domain vector (the whole vector, or one chunk
    // when the multi-GPU exchange is pipelined chunk by chunk against this kernel)
    // rows_per_part > 0: split-row walk (many rows of small blocks, where one workgroup per element tile would leave the
    // chip idle): workgroup row blockIdx.y sums its own rows in order into slab blockIdx.y of `part_out`; k_fold_parts
    // adds the slabs in part order afterwards (deterministic; not the bits of the single ordered sum)
    typedef typename vec_of<S, NS>::type V;
    if (rows_per_part > 0) {
        row0 += (int64_t)blockIdx.y * rows_per_part;
        if (row0 + rows_per_part < row1) row1 = row0 + rows_per_part;
        out = part_out + (int64_t)blockIdx.y * part_stride - s_begin;
        accumulate = 0;
    }
    const int64_t s0 = s_begin + ((int64_t)blockIdx.x * U * BLK + threadIdx.x) * NS;
    bool ok[U];
    V acc[U], mv[U];
    // clamp out-of-range vectors onto a valid address so the main loop is branch-free.  MIXED (round 5, session 3): these instantiations also serve rows
    // that are not whole, 16-byte aligned packs (see k_tall_diag_fwd): under-aligned accesses, the domain's last pack loaded from s_end - NS (st_pack)
    int64_t sk[U];
#pragma unroll
    for (int k = 0; k < U; k++) {
        ok[k] = (s0 + (int64_t)k * BLK * NS) < s_end;
        if constexpr (TAIL) {
            sk[k] = pack_start<NS>(ok[k] ? s0 + (int64_t)k * BLK * NS : s_begin, s_end);   // (idle lanes too: a range shorter than one pack ends with the vector, and a pack read from s_begin would run past it)
            acc[k] = (accumulate && ok[k]) ? ldu<false, S, NS>(out + sk[k]) : (V)(S)0;
            if (MODE == 1) mv[k] = ok[k] ? ldu<false, S, NS>(in + sk[k]) : (V)(S)0;
        } else {
            sk[k] = ok[k] ? s0 + (int64_t)k * BLK * NS : s_begin;
            acc[k] = (accumulate && ok[k]) ? ld<false>(reinterpret_cast<const V *>(out + s0 + (int64_t)k * BLK * NS)) : (V)(S)0;
            if (MODE == 1) mv[k] = ok[k] ? ld<false>(reinterpret_cast<const V *>(in + s0 + (int64_t)k * BLK * NS)) : (V)(S)0;
        }
    }

    int64_t i = row0;
    if constexpr (MIXED) {                 // rows of any elementwise kind (see k_tall_diag_fwd); zero blocks are skipped (1047)
        jh_dev_block blk[DEPTH], nxt[DEPTH];                               // the row table one batch ahead (scalar loads)
        if (i + DEPTH <= row1) {
#pragma unroll
            for (int j = 0; j < DEPTH; j++) nxt[j] = blocks[i + j];
        }
        for (; i + DEPTH <= row1; i += DEPTH) {
            V av[DEPTH][U], dv[DEPTH][U];
            const int64_t ahead = (i + 2 * DEPTH <= row1) ? i + DEPTH : i;
#pragma unroll
            for (int j = 0; j < DEPTH; j++) {
                blk[j] = nxt[j];
                nxt[j] = blocks[ahead + j];
            }
            // Round 6: a batch of PLAIN diagonals (what almost every batch of such an operator is -- [A; lambda I] has one special row) takes the all-diagonal
            // kernel's tight loop; the per-row kind switch below is what held these instantiations 20-30 % under the all-diagonal ones on rows of a few MiB
            // (one workgroup per CU: nothing hides the issue slots) -- fused A'A, one identity row: 256 x 2 MiB 4.9 -> 6.7 TB/s (profiles/exp_r06_chain_vs_mixed.txt)
            bool plain = true;
#pragma unroll
            for (int j = 0; j < DEPTH; j++) plain = plain && blk[j].kind == JH_OP_DIAG && !blk[j].adjoint;
            if (plain) {
#pragma unroll
                for (int j = 0; j < DEPTH; j++)
#pragma unroll
                    for (int k = 0; k < U; k++) {
                        av[j][k] = ldu<NT, S, NS>((const S *)blk[j].coeff + sk[k]);
                        if (MODE == 0) dv[j][k] = (V)in_scale * ldu<NT, S, NS>(in + (i + j) * n_scalars + sk[k]);
                    }
#pragma unroll
                for (int j = 0; j < DEPTH; j++)
#pragma unroll
                    for (int k = 0; k < U; k++) {
                        const V t = (MODE == 0) ? dv[j][k] : vmul<S, E, NS, V>(av[j][k], mv[k], false);
                        acc[k] = acc[k] + vmul<S, E, NS, V>(av[j][k], t, true);
                    }
                continue;
            }
#pragma unroll
            for (int j = 0; j < DEPTH; j++) {
                const bool on = blk[j].kind != JH_OP_ZERO, rc = block_reads_coeff(blk[j], false);
#pragma unroll
                for (int k = 0; k < U; k++) {
                    av[j][k] = rc ? ldu<NT, S, NS>((const S *)blk[j].coeff + sk[k]) : (V)(S)0;
                    dv[j][k] = (MODE == 0 && on) ? (V)in_scale * ldu<NT, S, NS>(in + (i + j) * n_scalars + sk[k]) : (V)(S)0;
                }
            }
#pragma unroll
            for (int j = 0; j < DEPTH; j++)
                if (blk[j].kind != JH_OP_ZERO) {
#pragma unroll
                    for (int k = 0; k < U; k++) {
                        const V t = (MODE == 0) ? dv[j][k] : apply_block_loaded<S, E, NS, V>(blk[j], mv[k], av[j][k], false, false);
                        acc[k] = acc[k] + apply_block_loaded<S, E, NS, V>(blk[j], t, av[j][k], true, false);   // _m .+= mul!(mtmp, op', _d)
                    }
                }
        }
        for (; i < row1; i++) {
            const jh_dev_block blk = blocks[i];
            if (blk.kind == JH_OP_ZERO) continue;
            const bool rc = block_reads_coeff(blk, false);
#pragma unroll
            for (int k = 0; k < U; k++) {
                const V c = rc ? ldu<NT, S, NS>((const S *)blk.coeff + sk[k]) : (V)(S)0;
                const V t = (MODE == 0) ? (V)in_scale * ldu<NT, S, NS>(in + i * n_scalars + sk[k]) : apply_block_loaded<S, E, NS, V>(blk, mv[k], c, false, false);
                acc[k] = acc[k] + apply_block_loaded<S, E, NS, V>(blk, t, c, true, false);
            }
        }
    }
    for (; !MIXED && !direct && i + DEPTH <= row1; i += DEPTH) {
        V av[DEPTH][U], dv[DEPTH][U];
#pragma unroll
        for (int j = 0; j < DEPTH; j++) {
            const S *a = a_base ? a_base + (i + j) * a_stride : (const S *)blocks[i + j].coeff;
#pragma unroll
            for (int k = 0; k < U; k++) {
                av[j][k] = ld<NT>(reinterpret_cast<const V *>(a + sk[k]));
                if (MODE == 0) dv[j][k] = ld<NT>(reinterpret_cast<const V *>(in + (i + j) * n_scalars + sk[k]));
            }
        }
#pragma unroll
        for (int j = 0; j < DEPTH; j++)
#pragma unroll
            for (int k = 0; k < U; k++) {
                V t = (MODE == 0) ? dv[j][k] : vmul<S, E, NS, V>(av[j][k], mv[k], false);   // d_i = a_i .* m   (1026)
                V p = vmul<S, E, NS, V>(av[j][k], t, true);                                 // mtmp = conj(a_i) .* d_i
                acc[k] = acc[k] + p;                                                         // _m .+= mtmp   (1049)
            }
    }
    for (; i < row1; i++) {
        const S *a = a_base ? a_base + i * a_stride : (const S *)blocks[i].coeff;
#pragma unroll
        for (int k = 0; k < U; k++) {
            V av = ld<NT>(reinterpret_cast<const V *>(a + sk[k]));
            V t = (MODE == 0) ? ld<NT>(reinterpret_cast<const V *>(in + i * n_scalars + sk[k])) : vmul<S, E, NS, V>(av, mv[k], false);
            V p = vmul<S, E, NS, V>(av, t, true);
            acc[k] = direct ? p : acc[k] + p;     // nrow == 1: mul!(_m, op', _d) writes directly (1051)
        }
    }
#pragma unroll
    for (int k = 0; k < U; k++)
        if (ok[k]) {
            if constexpr (TAIL) st_pack<false, S, NS>(out, s0 + (int64_t)k * BLK * NS, sk[k], acc[k]);
            else st<false>(reinterpret_cast<V *>(out + s0 + (int64_t)k * BLK * NS), acc[k]);
        }
}

// ---- the normal-equations pass of the device-resident CG loops (jh_lsqr.hip: cg_graph_impl; round 4) ---------------------------------
// ONE launch per iteration where the host-driven loop makes four (p <- s + bk p ; y = A'A p ; y += damp^2 p ; <p, y>): a thread owns one
// 16-byte pack of the domain -- it updates its pack of p (nobody else reads it in this launch: the rows below read coefficients only),
// walks all rows in order with DEPTH rows in flight exactly as k_tall_diag_adj MODE 1 does (product, product, add, each rounded: the bits
// of jh_blockop_normal_mul), adds the damping term with the lincomb's rounding, stores y and leaves its share of <p, y> (fp64) to the
// workgroup's partial.  Coefficients (bk, damp^2, the flags) come from device memory, so the launch is the same every iteration.
template <typename S, int E, int NS, int DEPTH, int BLK = 256, int U = 1, bool NT = true>
__global__ __launch_bounds__(BLK) void k_cg_normal(const jh_dev_block *__restrict__ blocks, int64_t nrow, const S *__restrict__ a_base, int64_t a_stride,
                                                   S *__restrict__ p, const S *__restrict__ sres, S *__restrict__ y, int64_t n_scalars,
                                                   const jh_cg_dev *__restrict__ stt, double *__restrict__ partials)
{
    typedef typename vec_of<S, NS>::type V;
    // a thread owns U packs, BLK packs apart (the shapes of the fused normal operator, launch_tall_adj_mixed: fat workgroups once the
    // blocks are big -- 64 x 128^3 with 256 x 1 x 8: 120 us per pass, with 512 x 2 x 2: 80)
    int64_t sk[U];
    int e0[U];
    bool ok[U];
    V pv[U], sv[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int64_t s0 = (((int64_t)blockIdx.x * U + u) * BLK + threadIdx.x) * NS;
        ok[u] = s0 < n_scalars;
        sk[u] = ok[u] ? pack_start<NS>(s0, n_scalars) : 0;                       // (rows need not be whole, 16-byte aligned packs: ldu / st_pack, the partial last pack's
        e0[u] = ok[u] ? (int)(s0 - sk[u]) : 0;                                   //  scalars before e0 belong to the neighbouring lane; round 5, last session)
    }
    // the state and this lane's packs of p and s are requested together, before the first decision (a launch of this size is paced by
    // round trips, not by bytes).  Folding the previous vector update's ||s||^2 partials and applying the second scalar update HERE, in
    // every workgroup (one more graph node less), was tried and lost: the whole grid then waits for a fold, a barrier and an fp64 chain
    // before its first coefficient load -- 28.7 us per iteration against 20.2 at 64 x 64^3 (profiles/bench_cgnr_sizes_r04.txt).
#pragma unroll
    for (int u = 0; u < U; u++) {
        pv[u] = ldu<false, S, NS>(p + sk[u]);
        sv[u] = ldu<false, S, NS>(sres + sk[u]);
    }
    const int done = stt->done, skip_p = stt->skip_p;
    const double bk = stt->bk, damp2 = stt->damp2;
#pragma unroll
    for (int u = 0; u < U; u++) asm volatile("" : "+v"(pv[u]), "+v"(sv[u]));
    if (done) return;
    if (!skip_p) {                                                           // p = 1*s + bk*p  (jh_lincomb's sequence: bk*p rounded, then the sum)
#pragma unroll
        for (int u = 0; u < U; u++) {
            const V bp = (V)(S)bk * pv[u];
            pv[u] = sv[u] + bp;
            if (ok[u]) st_pack<false, S, NS>(p, sk[u] + e0[u], sk[u], pv[u]);
        }
    }
    V acc[U];
#pragma unroll
    for (int u = 0; u < U; u++) acc[u] = (V)(S)0;
    int64_t i = 0;
    for (; i + DEPTH <= nrow; i += DEPTH) {
        V av[DEPTH][U];
#pragma unroll
        for (int j = 0; j < DEPTH; j++) {
            const S *a = a_base ? a_base + (i + j) * a_stride : (const S *)blocks[i + j].coeff;
#pragma unroll
            for (int u = 0; u < U; u++) av[j][u] = ldu<NT, S, NS>(a + sk[u]);
        }
#pragma unroll
        for (int j = 0; j < DEPTH; j++)
#pragma unroll
            for (int u = 0; u < U; u++) {
                const V t = vmul<S, E, NS, V>(av[j][u], pv[u], false);        // d_i = a_i .* p
                acc[u] = acc[u] + vmul<S, E, NS, V>(av[j][u], t, true);       // y .+= conj(a_i) .* d_i, rows in order
            }
    }
    for (; i < nrow; i++) {
        const S *a = a_base ? a_base + i * a_stride : (const S *)blocks[i].coeff;
#pragma unroll
        for (int u = 0; u < U; u++) {
            const V av = ldu<NT, S, NS>(a + sk[u]);
            const V t = vmul<S, E, NS, V>(av, pv[u], false);
            acc[u] = acc[u] + vmul<S, E, NS, V>(av, t, true);
        }
    }
    double part = 0.0;
#pragma unroll
    for (int u = 0; u < U; u++) {
        if (damp2 != 0.0) {                                                  // y = 1*y + damp^2*p
            const V dp = (V)(S)damp2 * pv[u];
            acc[u] = acc[u] + dp;
        }
        if (ok[u]) {
            st_pack<false, S, NS>(y, sk[u] + e0[u], sk[u], acc[u]);
#pragma unroll
            for (int e = 0; e < NS; e++) part += e >= e0[u] ? (double)pv[u][e] * (double)acc[u][e] : 0.0;   // Re <p, y>: over the scalars (a complex vector is 2n reals here)
        }
    }
    wg_sum_store<BLK>(part, partials + blockIdx.x);
}


// out[s] = sum over parts p = 0..nparts-1 (in that order within a part lane, part lanes in order) of parts[p][s - s_begin]:
// the second stage of the split-row walk.  64 vector lanes x 16 part lanes per workgroup; fp64 accumulation (exact
// conversions of S, so the fold adds no rounding of its own until the final cast); fixed order => deterministic.
// PL part lanes x (1024 / PL) vector lanes per workgroup: 16 x 64 for long domains; 64 x 16 (round 6) when the domain is short and the parts are many -- 262144 rows
// of 513 elements fold 2048 slabs of 129 packs: with sixteen part lanes that was three workgroups whose threads each walked 128 slabs one after the other
// (20.6 us under rocprofv3, a fifth of the fused A'A it finishes); sixty-four part lanes walk 32 (see profiles/rocprof_r06_thin_summary.md).
template <typename S, int NS, int PL = 16>
__global__ __launch_bounds__(1024) void k_fold_parts(const S *__restrict__ parts, int64_t part_stride, int nparts, S *__restrict__ out,
                                                     int64_t s_begin, int64_t s_end)
{
    typedef typename vec_of<S, NS>::type V;
    constexpr int VL = 1024 / PL;
    __shared__ double sm[PL][NS][VL];
    const int v = threadIdx.x % VL, q = threadIdx.x / VL;
    const int64_t s = s_begin + ((int64_t)blockIdx.x * VL + v) * NS;
    const bool ok = s < s_end;
    const int64_t sc = ok ? pack_start<NS>(s, s_end) : s_begin;     // (the last, partial pack of a domain that is not whole packs: loaded from s_end - NS, st_pack)
    double acc[NS];
#pragma unroll
    for (int e = 0; e < NS; e++) acc[e] = 0.0;
    if (ok) {
        const S *src = parts + (sc - s_begin);
#pragma unroll 4
        for (int p = q; p < nparts; p += PL) {
            const V x = ldu<false, S, NS>(src + (int64_t)p * part_stride);
#pragma unroll
            for (int e = 0; e < NS; e++) acc[e] += (double)x[e];
        }
    }
#pragma unroll
    for (int e = 0; e < NS; e++) sm[q][e][v] = acc[e];
    __syncthreads();
    if (q == 0 && ok) {
        V r;
#pragma unroll
        for (int e = 0; e < NS; e++) {
            double t = acc[e];
#pragma unroll 8
            for (int qq = 1; qq < PL; qq++) t += sm[qq][e][v];
            r[e] = (S)t;
        }
        st_pack<false, S, NS>(out, s, sc, r);
    }
}

// ------------------------------------------------------------------ launch helpers ------------
// Kernel shapes, fitted to interleaved sweeps on MI355X (profiles/sweep_r01*.txt, profiles/repeat_r01.txt; Float32):
//   forward  1024 x 256^3 (128 GiB): sequential row sweep, 1024 threads x 8 vectors x 16 rows: 6.04 TB/s in
//            every process.  Walking all rows concurrently (order 1) reaches 6.4-6.5 TB/s in some processes
//            and 5.2-5.6 TB/s in others (same binary, same box: physical placement luck), banded walks sit
//            in between -- so the sequential sweep is the default and the other orders stay behind the knob.
//            256 x 256^3 ... 16 x 256^3: 256 threads x 4 vectors x 4 rows (5.8-6.3 TB/s)
//            64 x 128^3, 1024 x 64^3 (1-2 GiB): 256 threads x 1 vector x 2 rows (6.3-6.4 TB/s)
//   adjoint  wants FEW, FAT workgroups: 4 vectors per thread as long as >= 256 workgroups remain
//            (1024 x 256^3: 6.6-6.7 TB/s; 64 x 128^3: 6.8-7.1 TB/s; 1024 x 64^3: 1 vector, 7.1 TB/s)
//   fused A'A reads one stream, so it keeps twice the rows in flight.

TallShape pick_fwd_shape(int64_t nvec, int64_t nrow, size_t vec_bytes)
{
    jh_context &c = jh_ctx();
    (void)vec_bytes;
    TallShape s;
    if (nvec >= ((int64_t)1 << 21) && nrow >= 512) s = TallShape{1024, 8, 16, 0};
    else if (nvec >= ((int64_t)1 << 21)) s = TallShape{256, 1, 1, 1, 32};   // blocks of >= 32 MiB, fewer than 512 rows: column bands (16-32 x 256^3: 6.05-6.19 -> 6.67 TB/s with
                                                                            // 256 x 4 x 4 rows sequential before; profiles/exp_r04_small_fwd.txt)
    else s = TallShape{256, 1, 2, 0};
    if (c.fwd_wg) s.wg = (int)c.fwd_wg;
    if (c.fwd_unroll) s.unroll = (int)c.fwd_unroll;
    if (c.fwd_group) s.aux = (int)c.fwd_group;
    if (c.fwd_order >= 0) s.order = (int)c.fwd_order;
    if (c.fwd_ctiles >= 0) s.ctiles = (int)c.fwd_ctiles;
    return s;
}

}  // namespace
namespace jhb {
// the shapes of the all-diagonal ordered walk that exist: what the size rule of pick_adj_shape selects without knobs
//   adjoint  (mode 0): 256 x {1x4, 2x4, 4x2, 4x4}, 512 x 4x4 (>= 4 M packs, >= 512 rows), 1024 x 4x2 (>= 4 M packs, fewer rows)
//   fused A'A (mode 1): 256 x {1x8, 2x8, 4x4}, 1024 x 4x4 (>= 4 M packs)
constexpr bool adj_shape_built(int mode, int wg, int u, int d)
{
    if (mode == 0)
        return (wg == 256 && ((u == 1 && d == 4) || (u == 2 && d == 4) || (u == 4 && (d == 2 || d == 4)))) || (wg == 512 && u == 4 && d == 4) ||
               (wg == 1024 && u == 4 && d == 2);
    return (wg == 256 && ((u == 1 && d == 8) || (u == 2 && d == 8) || (u == 4 && d == 4))) || (wg == 1024 && u == 4 && d == 4);
}
TallShape pick_adj_shape(int64_t nvec, int64_t nrow, int mode)
{
    jh_context &c = jh_ctx();
    TallShape s{256, 1, 4, 0};
    if (nvec >= 4 * 256 * 256) s.unroll = 4;
    else if (nvec >= 2 * 256 * 256) s.unroll = 2;
    if (s.unroll == 4) s.aux = (nrow >= 256) ? 4 : 2;
    if (nvec >= ((int64_t)1 << 22)) s.wg = 512;
    if (mode == 0 && nvec >= ((int64_t)1 << 22) && nrow < 512) { s.wg = 1024; s.aux = 2; }   // 128..256 x 256^3: +2..7 % (sweep_r01_pair_*)
    if (mode == 1) {                                   // fused normal operator: one input stream
        s.aux = (s.unroll == 4) ? 4 : 8;
        if (nvec >= ((int64_t)1 << 22)) s.wg = 1024;
    }
    if (c.adj_wg) s.wg = (int)c.adj_wg;
    if (c.adj_unroll) s.unroll = (int)c.adj_unroll;
    if (c.adj_depth) s.aux = (int)c.adj_depth;
    if (s.unroll != 1 && s.unroll != 2 && s.unroll != 4) s.unroll = s.unroll > 2 ? 4 : (s.unroll > 1 ? 2 : 1);
    if (s.unroll == 4) s.aux = s.aux >= 4 ? 4 : 2;     // instantiated: 4 x {2, 4} (4 x 8: register budget), {1, 2} x {4, 8} -- a knob pair
    else s.aux = s.aux >= 8 ? 8 : 4;                   // outside that list runs the nearest shape of it (same bits)
    // a 1024-thread workgroup has 128 VGPRs per lane: the two-stream adjoint keeps at most 8 packs per stream in flight there
    // (2 x 8 and 4 x 4 spilled 108-176 bytes per lane to scratch; same bits with fewer rows in flight)
    if (mode == 0 && s.wg == 1024 && s.unroll * s.aux > 8) s.aux = 8 / s.unroll;
    // Round 6: only the (workgroup, packs, rows) shapes the SIZE RULE above can select are compiled (adj_shape_built: ten per element type, load policy aside,
    // where rounds 1-5 compiled the knobs' whole 3 x 6 matrix -- 192 instantiations no default route reached).  A knob triple that names another shape runs
    // the built shape of its workgroup size with the nearest packs x rows, else the rule's own shape (same bits: the rows are summed in order whatever the shape).
    if (!adj_shape_built(mode, s.wg, s.unroll, s.aux)) {
        static const int pairs[6][2] = {{1, 4}, {1, 8}, {2, 4}, {2, 8}, {4, 2}, {4, 4}};
        int best = -1, best_d = 1 << 30;
        for (int k = 0; k < 6; k++)
            if (adj_shape_built(mode, s.wg, pairs[k][0], pairs[k][1])) {
                const int d = 16 * (pairs[k][0] > s.unroll ? pairs[k][0] - s.unroll : s.unroll - pairs[k][0]) + (pairs[k][1] > s.aux ? pairs[k][1] - s.aux : s.aux - pairs[k][1]);
                if (d < best_d) { best_d = d; best = k; }
            }
        if (best >= 0) { s.unroll = pairs[best][0]; s.aux = pairs[best][1]; }
        else { s.wg = (mode == 1 && s.wg == 512) ? 1024 : 256; s.unroll = 4; s.aux = 4; }       // (the fused A'A has no 512-lane shape)
    }
    return s;
}
}  // namespace jhb
namespace {

// Split-row walk of the adjoint-shaped kernels.  The ordered walk gives one thread a 16-byte vector of the DOMAIN and
// all rows: with n elements per block that is n/4 threads, so a tall operator of many SMALL blocks (seismic traces
// rather than volumes) leaves most of the chip idle -- 1 GiB of 4096-element Float32 rows: 10.7 ms, 200 GB/s
// (profiles/exp_r01_small_blocks.txt).  When the ordered walk would launch fewer workgroups than the chip has CUs, the
// rows are cut into `parts` contiguous ranges, workgroup row y sums range y in order into its own slab, and k_fold_parts
// adds the slabs in a fixed order.  Deterministic, but not the bits of the single ordered sum (tolerance parity, like the
// multi-GPU sum).  Knob adj_split: -1 automatic, 0 never (always the ordered, bit-exact walk), k > 1 that many parts.
}  // namespace
namespace jhb {
int64_t pick_adj_parts(int64_t gx, int64_t nrow)
{
    jh_context &c = jh_ctx();
    if (c.adj_split == 0 || nrow < 4) return 1;
    int64_t parts;
    if (c.adj_split > 0) parts = c.adj_split;
    else {
        if (gx >= c.cu_count || nrow < 256) return 1;                     // small operators keep the ordered, bit-exact walk
        parts = (8 * (int64_t)c.cu_count + gx - 1) / gx;                 // ~8 workgroups per CU
        if (parts > nrow / 16) parts = nrow / 16;                       // at least 16 rows per part
    }
    if (parts > nrow / 2) parts = nrow / 2;
    if (parts > 65535) parts = 65535;                                    // gridDim.y
    return parts < 2 ? 1 : parts;
}
}  // namespace jhb
namespace {

// The split walk's slabs live at the start of the context's scratch buffer.  A caller may have placed the OUTPUT behind slabs it reserved there
// (split_adjoint_tmp): the walk must then neither regrow the buffer (that frees what `out` points into) nor need more slabs than were reserved
// (they would lie over `out`).  Either would be silent corruption, so it is an error here.
static int adj_slabs(void *out, size_t bytes, void **slabs)
{
    jh_context &c = jh_ctx();
    const char *b = (const char *)c.scratch_dev, *o = (const char *)out;
    if (b && o >= b && o < b + c.scratch_cap)
        JH_REQUIRE(bytes <= c.scratch_cap && o >= b + bytes, "split adjoint: %zu bytes of slabs would overlap the output reserved %zu bytes into the scratch buffer",
                   bytes, (size_t)(o - b));
    return jh_ensure_scratch(bytes, slabs);
}

}  // namespace
namespace jhb {
int split_slabs(void *out, size_t bytes, void **slabs) { return adj_slabs(out, bytes, slabs); }
}  // namespace jhb
namespace {

template <typename S, int NS>
int launch_fold_parts(const void *parts, int64_t part_stride, int64_t nparts, void *out, int64_t s_begin, int64_t s_end)
{
    jh_context &c = jh_ctx();
    const int64_t packs = (s_end - s_begin + NS - 1) / NS;
    if (packs <= 64 * 8 && nparts >= 256) {                              // a short domain under many slabs: more part lanes, fewer slabs per thread
        hipLaunchKernelGGL((k_fold_parts<S, NS, 64>), dim3((unsigned)((packs + 15) / 16)), dim3(1024), 0, c.stream, (const S *)parts, part_stride, (int)nparts,
                           (S *)out, s_begin, s_end);
        JH_CHECK_HIP(hipGetLastError());
        return JH_OK;
    }
    const int64_t gx = (packs + 63) / 64;
    hipLaunchKernelGGL((k_fold_parts<S, NS>), dim3((unsigned)gx), dim3(1024), 0, c.stream, (const S *)parts, part_stride, (int)nparts,
                       (S *)out, s_begin, s_end);
    JH_CHECK_HIP(hipGetLastError());
    return JH_OK;
}

template <typename S, int E, int NS, bool NT, int BLK>
int launch_tall_fwd_u(const jh_blockop *op, void *d, const void *m, int64_t n_scalars, const TallShape &sh)
{
    jh_context &c = jh_ctx();
    const S *a_base = op->diag_strided ? (const S *)op->blocks[0].coeff : nullptr;
    const int64_t a_stride = op->diag_stride_elems * E;
    int64_t G = sh.aux;
    if (G > op->nrow) G = op->nrow;
    int64_t gy = (op->nrow + G - 1) / G;
    {   // HIP: grid x block must stay below 2^32 threads
        const int64_t gx0 = (n_scalars / NS + (int64_t)sh.unroll * BLK - 1) / ((int64_t)sh.unroll * BLK);
        while (gx0 * gy * BLK >= ((int64_t)1 << 32) && G < op->nrow) { G *= 2; gy = (op->nrow + G - 1) / G; }
    }
    int64_t band = sh.order <= 0 ? 1 : (sh.order == 1 ? gy : sh.order);   // order: 0 sequential, 1 all rows, k>1 = k groups per band
    if (band > gy) band = gy;
    c.last_fwd_walk = sh.ctiles ? 2 : sh.order;
    c.last_fwd_rows_per_wg = G;
#define JH_FWD_CASE(U)                                                                                               \
    case U: {                                                                                                         \
        int64_t gx = (n_scalars + (int64_t)U * BLK * NS - 1) / ((int64_t)U * BLK * NS);                               \
        JH_REQUIRE(gx * gy * BLK < (int64_t)1 << 32, "tall forward: grid of %lld workgroups is too large", (long long)(gx * gy)); \
        hipLaunchKernelGGL((k_tall_diag_fwd<S, E, NS, U, NT, BLK>), dim3((unsigned)(gx * gy)), dim3(BLK), 0, c.stream, \
                           op->dev_blocks, op->nrow, (int)G, a_base, a_stride, (const S *)m, (S *)d, n_scalars,      \
                           (unsigned)gx, (unsigned)gy, (unsigned)band, (unsigned)sh.ctiles);                       \
    } break;
    switch (sh.unroll) {
        JH_FWD_CASE(1)
        JH_FWD_CASE(2)
        JH_FWD_CASE(4)
        JH_FWD_CASE(8)
    default: return jh_fail(JH_ERR_INVALID, "fwd_unroll %d unsupported", sh.unroll);
    }
#undef JH_FWD_CASE
    JH_CHECK_HIP(hipGetLastError());
    return JH_OK;
}

template <typename S, int E, int NS, bool NT, int MODE, int BLK>
int launch_tall_adj_u(const jh_blockop *op, void *out, const void *in, int64_t n_scalars, const TallShape &sh, int64_t s_begin,
                      int64_t s_end)
{
    jh_context &c = jh_ctx();
    const S *a_base = op->diag_strided ? (const S *)op->blocks[0].coeff : nullptr;
    const int64_t a_stride = op->diag_stride_elems * E;
    const int direct = (op->nrow == 1 && MODE == 0) ? 1 : 0;
    // A 128 GiB walk runs 2-3 % faster as two launches over 512 rows each than as one (profiles/exp_r01_adj_row_chunks.txt,
    // exp_r01_adj_by_rows.txt); the second launch continues the ordered sum, so the bits do not change.
    int64_t rows_per_launch = op->nrow;
    if (c.adj_rows_per_launch > 0) rows_per_launch = c.adj_rows_per_launch < op->nrow ? c.adj_rows_per_launch : op->nrow;
    else if (op->nrow >= 768 && (double)op->nrow * (double)n_scalars * sizeof(S) >= 48.0 * (double)(1ull << 30)) rows_per_launch = 512;
    c.last_adj_launches = (op->nrow + rows_per_launch - 1) / rows_per_launch;
    // many rows of small blocks: split-row walk (pick_adj_parts) -- one launch over (tiles, parts), then the fold
    const int64_t gx0 = (s_end - s_begin + (int64_t)sh.unroll * BLK * NS - 1) / ((int64_t)sh.unroll * BLK * NS);
    const int from_found = (MODE == 0) ? c.adj_from_found : 0;              // continue from what `out` holds (a wide operator's forward)
    int64_t parts = (direct || from_found) ? 1 : pick_adj_parts(gx0, op->nrow);
    int64_t rows_per_part = 0;
    const int64_t part_stride = s_end - s_begin;
    void *slabs = nullptr;
    if (parts > 1) {
        rows_per_part = (op->nrow + parts - 1) / parts;
        parts = (op->nrow + rows_per_part - 1) / rows_per_part;                // no empty part
        JH_TRY(adj_slabs(out, (size_t)parts * (size_t)part_stride * sizeof(S), &slabs));
        rows_per_launch = op->nrow;
        c.last_adj_launches = 1;
    }
    c.last_adj_parts = parts;
#define JH_ADJ_CASE(U, DEPTH)                                                                                          \
    if constexpr (jhb::adj_shape_built(MODE, BLK, U, DEPTH))                                                           \
    if (sh.unroll == U && sh.aux == DEPTH) {                                                                           \
        int64_t gx = (s_end - s_begin + (int64_t)U * BLK * NS - 1) / ((int64_t)U * BLK * NS);                          \
        for (int64_t r0 = 0; r0 < op->nrow; r0 += rows_per_launch) {                                                   \
            const int64_t r1 = r0 + rows_per_launch < op->nrow ? r0 + rows_per_launch : op->nrow;                        \
            hipLaunchKernelGGL((k_tall_diag_adj<S, E, NS, U, DEPTH, NT, MODE, BLK>), dim3((unsigned)gx, (unsigned)parts), \
                               dim3(BLK), 0,                                                                           \
                               c.stream, op->dev_blocks, op->nrow, a_base, a_stride, (S *)out, (const S *)in, n_scalars,   \
                               direct, s_begin, s_end, r0, r1, (r0 > 0 || from_found) ? 1 : 0, rows_per_part, (S *)slabs, part_stride); \
            JH_CHECK_HIP(hipGetLastError());                                                                           \
        }                                                                                                              \
        if (parts > 1) return launch_fold_parts<S, NS>(slabs, part_stride, parts, out, s_begin, s_end);                \
        return JH_OK;                                                                                                  \
    }
    // (round 6: of this list only the combinations with a workgroup size that the size rule can select -- adj_shape_built)
    // Round 5: the six (packs per lane, rows in flight) shapes the size rule of pick_adj_shape can select.  Five more used to be compiled
    // for the knobs alone -- (1,1) (1,2) (2,1) (2,2) (4,1): 240 of this kernel's 524 instantiations, the bulk of the library's build
    // time -- and lost every sweep of rounds 1-2 (profiles/sweep_r02_adj_rows.txt); pick_adj_shape now maps a knob pair to the nearest
    // shape of this list (same bits: the rows are summed in order whatever the shape).
    JH_ADJ_CASE(1, 4) JH_ADJ_CASE(1, 8)
    JH_ADJ_CASE(2, 4) JH_ADJ_CASE(2, 8)
    JH_ADJ_CASE(4, 2) JH_ADJ_CASE(4, 4)
#undef JH_ADJ_CASE
    return jh_fail(JH_ERR_INVALID, "adj_unroll %d x adj_depth %d unsupported", sh.unroll, sh.aux);
}

template <typename S, int E, int NS>
int launch_tall_fwd_shape(const jh_blockop *op, void *d, const void *m, int64_t n_scalars, const TallShape &sh);

// the shapes the first forward of a large operator is timed with: which one wins differs from process to process
// (profiles/repeat_r01*.txt, sweep_r01_order.txt): sequential sweeps, and walks that touch every row group concurrently
// -- and the column-persistent walk (rows per workgroup = all rows: a workgroup keeps its m tile and streams every block row
// through it, no m re-reads), which is the best of the placement-independent shapes (22.0 vs 23.0 ms for the 16-row sweep,
// profiles/sweep_r01_fwd_persistent_1024x256.txt)
// Candidates 6 and 7 (late round 2) are ONE block row per workgroup with all rows concurrent -- workgroups that are born, move
// one tile of one row and die: the fastest shapes at the row counts a rank owns on 2 and 4 GPUs (512 rows: 6.32 TB/s against
// 5.98 for the best of the first six, 256 rows: 6.2 against 6.06; profiles/sweep_r02_fwd_rows.txt), equal to the others at 1024.
// Candidates 8 and 9 (late round 4): one block row per workgroup in COLUMN bands of 32 / 64 tiles (k_tall_diag_fwd's ctiles decode) -- tried by
// operators of fewer than 1024 rows only (the row counts a rank owns on 2 / 4 / 8 GPUs): 128 rows 5.8 -> 6.4 TB/s, 256 rows 5.8 -> 6.1, 512 rows
// 6.05 -> 6.4 where a copy between the same slabs runs at 6.5; at 1024 rows the row-concurrent walk equals the copy and the bands lose 2 %.
constexpr int K_FWD_CANDIDATES = 10, K_FWD_CANDIDATES_TALL = 8;          // (operators of >= 1024 rows of blocks >= 64 MiB try the first eight)
static_assert(2 * K_FWD_CANDIDATES + 4 <= jh_blockop::LazyTune::SLOTS, "two passes per candidate and the play-off must fit the trial slots");
static_assert(K_FWD_CANDIDATES <= jh_blockop::LazyTune::MAXC, "the candidates' records");
const TallShape k_fwd_candidates[K_FWD_CANDIDATES] = {TallShape{1024, 8, 16, 0}, TallShape{512, 1, 2, 1}, TallShape{256, 4, 4, 0},
                                                      TallShape{256, 4, 16, 1}, TallShape{512, 4, 8, 1}, TallShape{1024, 8, 1 << 20, 0},
                                                      TallShape{512, 8, 1, 1},  TallShape{256, 1, 1, 1},
                                                      TallShape{256, 1, 1, 1, 32}, TallShape{256, 1, 1, 1, 64}};

// the order in which an untuned operator tries them: the one-row-per-workgroup walks first (the winners on most boxes and pairings,
// profiles/repeat_r03_boxes.txt), the sequential sweeps last
const int k_fwd_trial_order[K_FWD_CANDIDATES_TALL] = {7, 6, 1, 4, 3, 2, 0, 5};
const int k_fwd_trial_order_few[K_FWD_CANDIDATES] = {8, 9, 7, 6, 1, 4, 3, 2, 0, 5};
// (the bands also for >= 1024 rows of blocks below 64 MiB: 1024 x 128^3 with a non-diagonal row runs its banded forward + adjoint pair at 6.55 TB/s
// where the all-diagonal operator's best of eight gave 6.31)
// The shape candidate k RUNS on rows of `nvec` 16-byte packs -- in a trial, once chosen, in the periodic re-check and when an operator inherits
// the choice (walk memory): the column-persistent walk (candidate 5) has one workgroup per 128 KiB of a ROW, so with small blocks it is a
// handful of workgroups walking thousands of rows (4096 x 64^3: 17.8 ms where the others take 1.4-1.7) -- there candidate 5 IS candidate 0's
// shape, everywhere, so a timing of "5" is always a timing of what a choice of 5 would run (round-4 advisor finding: the trial alone was
// substituted, and a tie or play-off won by 5 then ran the real column-persistent walk for ~192 calls until the re-check rotated it out)
static inline TallShape fwd_candidate_shape(int k, int64_t nvec)
{
    if (k == 5 && nvec < (int64_t)512 * 1024 * 8) return k_fwd_candidates[0];
    return k_fwd_candidates[k];
}

static inline int fwd_candidates_of(const jh_blockop *op)
{
    const bool small_blocks = (double)op->row_len[0] * (double)jh_dtype_size(op->dtype) < (double)(64u << 20);
    return (op->nrow < 1024 || small_blocks) ? K_FWD_CANDIDATES : K_FWD_CANDIDATES_TALL;
}

// For operators far larger than the caches the row-concurrent walk is 5-7 % faster than the sequential sweep in some
// processes and 10-15 % slower in others (profiles/repeat_r01.txt: same binary, same box; it depends on where the slabs landed
// physically), so the shape is chosen by measurement -- LAZILY: while an operator is untuned, each real forward call runs the
// next candidate shape between two events (every candidate computes the same bits), nothing is launched that the caller did
// not ask for and the host never waits; a later call harvests the finished timings with hipEventQuery and, once every
// candidate has been measured twice (the first pass also warms caches and TLBs), keeps the fastest.  jh_blockop_mul returns
// after enqueue, always.  Skipped while the stream is being captured.  jh_blockop_tune_get/set export / import the choice.
}  // namespace
namespace jhb {
void lazy_release(jh_blockop::LazyTune &t)
{
    for (auto &pair : t.ev)
        for (auto &e : pair)
            if (e) { (void)hipEventDestroy(e); e = nullptr; }
    for (auto &e : t.rc_ev)                                                 // (the re-check's pair exists only after a choice was made)
        if (e) { (void)hipEventDestroy(e); e = nullptr; }
    t.rc_in_flight = false;
}
}  // namespace jhb
namespace {

}  // namespace
namespace jhb {
void lazy_reset(jh_blockop::LazyTune &t)
{
    lazy_release(t);
    for (auto &st : t.state) st = 0;
    for (auto &m : t.ms) m = 0.f;
    for (auto &m : t.best_ms) m = 0.f;
    t.launched = 0;
    t.playoff[0] = t.playoff[1] = -1;
    t.calls = 0;
    t.rc_in_flight = false;
    t.rc_slow = 0;
}
}  // namespace jhb
namespace {

// Which candidate should THIS call run?  Trial slots are laid out pass-major after `warm` untimed-in-effect slots (their
// timings are discarded): slot = warm + pass * ncand + candidate.  Returns the candidate and, when the call is a trial, its
// slot (else -1).  Once every slot has a timing, *choice = the candidate with the best time over its passes -- candidate 0
// unless another one beats it by `margin` -- and the events are released.
// playoff (round 3): two timings per candidate decide between shapes that often differ by 1-2 %, less than the timings scatter; so
// when the runner-up is within 3 % of the winner the two run a play-off -- four more of the caller's own calls, alternating
// winner / runner-up / winner / runner-up, each between two events like the trials -- and the best time over ALL of a
// candidate's samples decides.
// `order` (optional, ncand entries): the candidate that trial j of a pass runs -- the likely winners first, so that an operator that
// lives for a handful of calls only (a caller in the reference's style builds operators all the time) spends them on good shapes
}  // namespace
namespace jhb {
int lazy_next(jh_blockop::LazyTune &t, int ncand, int npass, int warm, float margin, int *choice, int *slot, bool playoff,
              const int *order)
{
    *slot = -1;
    const int regular = warm + ncand * npass;
    const int total = regular + (t.playoff[0] >= 0 ? 4 : 0);
    int measured = 0;
    for (int k = 0; k < t.launched; k++) {                                  // harvest what has finished (non-blocking)
        if (t.state[k] == 1 && hipEventQuery(t.ev[k][1]) == hipSuccess) {
            float ms = 0.f;
            t.state[k] = (hipEventElapsedTime(&ms, t.ev[k][0], t.ev[k][1]) == hipSuccess && ms > 0.f) ? 2 : 3;
            t.ms[k] = ms;
        }
        if (t.state[k] >= 2) measured++;
    }
    (void)hipGetLastError();                                                // hipEventQuery's hipErrorNotReady is not an error
    if (measured == total) {
        float best[jh_blockop::LazyTune::MAXC] = {};
        auto take = [&](int c, int k) { if (t.state[k] == 2 && (best[c] == 0.f || t.ms[k] < best[c])) best[c] = t.ms[k]; };
        for (int j = 0; j < ncand; j++)
            for (int p = 0; p < npass; p++) take(order ? order[j] : j, warm + p * ncand + j);
        if (t.playoff[0] >= 0)
            for (int k = 0; k < 4; k++) take(t.playoff[k & 1], regular + k);
        int pick = 0, runner = -1;
        for (int c = 1; c < ncand; c++)
            if (best[c] > 0.f && (best[pick] == 0.f || best[c] < (1.f - margin) * best[pick])) pick = c;
        for (int c = 0; c < ncand; c++)
            if (c != pick && best[c] > 0.f && (runner < 0 || best[c] < best[runner])) runner = c;
        if (playoff && t.playoff[0] < 0 && runner >= 0 && best[pick] > 0.f && best[runner] <= 1.03f * best[pick] &&
            regular + 4 <= jh_blockop::LazyTune::SLOTS) {
            t.playoff[0] = pick;                                            // four more trials; the choice waits for them
            t.playoff[1] = runner;
            *slot = t.launched;
            return pick;
        }
        for (int c = 0; c < ncand && c < jh_blockop::LazyTune::MAXC; c++) t.best_ms[c] = best[c];
        *choice = pick;
        lazy_release(t);
        return pick;
    }
    if (t.launched < total) {
        *slot = t.launched;
        if (*slot >= regular) return t.playoff[(*slot - regular) & 1];
        if (*slot < warm) return order ? order[0] : 0;
        return order ? order[(*slot - warm) % ncand] : (*slot - warm) % ncand;
    }
    return t.playoff[0] >= 0 ? t.playoff[0] : (order ? order[0] : 0);       // every trial is in flight: the (provisional) default meanwhile
}
}  // namespace jhb
namespace {

// Periodic re-check of a choice already made (round 3): every 64th call of the chosen shape is timed between two events (again the
// caller's own launch, harvested later without waiting).  Three such samples in a row that are more than 3 % slower than the best
// time ANOTHER candidate recorded during the search rotate that candidate in; the dethroned one's record is replaced by what it
// has just shown, so the two cannot flip back and forth on stale numbers.  Returns true when THIS call should be timed.
bool recheck_should_time(jh_blockop::LazyTune &t, int ncand, int *choice)
{
    if (t.rc_in_flight && hipEventQuery(t.rc_ev[1]) == hipSuccess) {
        float ms = 0.f;
        t.rc_in_flight = false;
        if (hipEventElapsedTime(&ms, t.rc_ev[0], t.rc_ev[1]) == hipSuccess && ms > 0.f && *choice >= 0 && *choice < jh_blockop::LazyTune::MAXC) {
            int other = -1;
            for (int c = 0; c < ncand && c < jh_blockop::LazyTune::MAXC; c++)
                if (c != *choice && t.best_ms[c] > 0.f && (other < 0 || t.best_ms[c] < t.best_ms[other])) other = c;
            if (other >= 0 && ms > 1.03f * t.best_ms[other]) {
                if (++t.rc_slow >= 3) {
                    t.best_ms[*choice] = ms;
                    *choice = other;
                    t.rc_slow = 0;
                    t.switches++;
                }
            } else {
                t.rc_slow = 0;
                if (ms < t.best_ms[*choice] || t.best_ms[*choice] == 0.f) t.best_ms[*choice] = ms;
            }
        }
    }
    (void)hipGetLastError();
    t.calls++;
    // every 64th call -- and the three calls after a slow sample, so that a real slowdown is confirmed (or dismissed) at once
    return !t.rc_in_flight && (t.calls % 64 == 0 || t.rc_slow > 0);
}

bool recheck_begin(jh_blockop::LazyTune &t, hipStream_t st)
{
    for (auto &e : t.rc_ev)
        if (!e && hipEventCreate(&e) != hipSuccess) return false;
    return hipEventRecord(t.rc_ev[0], st) == hipSuccess;
}

void recheck_end(jh_blockop::LazyTune &t, hipStream_t st, bool ok) { t.rc_in_flight = ok && hipEventRecord(t.rc_ev[1], st) == hipSuccess; }

}  // namespace
namespace jhb {
bool lazy_begin(jh_blockop::LazyTune &t, int slot, hipStream_t st)
{
    return hipEventCreate(&t.ev[slot][0]) == hipSuccess && hipEventCreate(&t.ev[slot][1]) == hipSuccess &&
           hipEventRecord(t.ev[slot][0], st) == hipSuccess;
}
}  // namespace jhb
namespace {

}  // namespace
namespace jhb {
void lazy_end(jh_blockop::LazyTune &t, int slot, hipStream_t st, bool ok)
{
    ok = ok && hipEventRecord(t.ev[slot][1], st) == hipSuccess;
    t.state[slot] = ok ? 1 : 3;
    t.launched = slot + 1;
}
}  // namespace jhb
namespace {

}  // namespace
namespace jhb {
bool stream_is_capturing(hipStream_t st)
{
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    return !(hipStreamIsCapturing(st, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone);
}
}  // namespace jhb
namespace {

// Round 4: what an operator of this SHAPE chose last time.  A caller in the reference's style builds operators again and again (a new
// JopBlock per outer iteration, per shot set): each would spend its first 16-20 forwards on the candidate walks, the slow ones included --
// the first `d = A*m` of a new operator ran the 16-row sweep (23-24 ms at the headline size) whatever the last operator had found.
// A new operator of a shape (device, element type, rows, block length, strided or table addressing) seen before starts with that
// choice and its records; the periodic re-check (every 64th call) still corrects it.  Knob walk_memory (0: every operator measures);
// jh_blockop_tune_set(op, "fwd_walk", -1) makes that operator measure for itself.
struct WalkKey {
    int device, dtype, strided;
    int64_t nrow, n_scalars;
    bool operator<(const WalkKey &o) const
    {
        return std::tie(device, dtype, strided, nrow, n_scalars) < std::tie(o.device, o.dtype, o.strided, o.nrow, o.n_scalars);
    }
};
struct WalkRecord { int walk; float best_ms[jh_blockop::LazyTune::MAXC]; };
std::mutex g_walk_mutex;
std::map<WalkKey, WalkRecord> g_walk_memory;

WalkKey walk_key(const jh_blockop *op, int64_t n_scalars) { return WalkKey{jh_ctx().device, op->dtype, op->diag_strided ? 1 : 0, op->nrow, n_scalars}; }

void walk_remember(const jh_blockop *op, int64_t n_scalars)
{
    if (op->fwd_walk < 0 || op->fwd_walk >= K_FWD_CANDIDATES) return;
    WalkRecord r{op->fwd_walk, {}};
    for (int k = 0; k < jh_blockop::LazyTune::MAXC; k++) r.best_ms[k] = op->fwd_tune.best_ms[k];
    std::lock_guard<std::mutex> lock(g_walk_mutex);
    g_walk_memory[walk_key(op, n_scalars)] = r;
}

bool walk_recall(const jh_blockop *op, int64_t n_scalars)
{
    std::lock_guard<std::mutex> lock(g_walk_mutex);
    auto it = g_walk_memory.find(walk_key(op, n_scalars));
    if (it == g_walk_memory.end()) return false;
    op->fwd_walk = it->second.walk;
    for (int k = 0; k < jh_blockop::LazyTune::MAXC; k++) op->fwd_tune.best_ms[k] = it->second.best_ms[k];
    op->walk_inherited = true;
    return true;
}

template <typename S, int E, int NS>
int launch_tall_fwd(const jh_blockop *op, void *d, const void *m, int64_t n_scalars)
{
    jh_context &c = jh_ctx();
    TallShape sh = pick_fwd_shape(n_scalars / NS, op->nrow, sizeof(S) * NS);
    const bool knobs_free = !c.fwd_wg && !c.fwd_unroll && !c.fwd_group && c.fwd_order < 0;
    const double stream_bytes = 2.0 * (double)op->nrow * (double)n_scalars * sizeof(S);
    if (c.autotune && knobs_free && stream_bytes >= 8.0 * (double)(1ull << 30) && op->nrow >= 64) {
        int slot = -1;
        if (op->fwd_walk < 0 && op->fwd_tune.launched == 0 && !op->walk_measure_again && c.walk_memory) (void)walk_recall(op, n_scalars);
        if (op->fwd_walk < 0) {
            if (!stream_is_capturing(c.stream)) {
                const int nc = fwd_candidates_of(op);
                const int k = lazy_next(op->fwd_tune, nc, 2, 0, 0.f, &op->fwd_walk, &slot, true, nc == K_FWD_CANDIDATES ? k_fwd_trial_order_few : k_fwd_trial_order);
                if (k >= 0 && k < K_FWD_CANDIDATES) sh = fwd_candidate_shape(k, n_scalars / NS);
                if (op->fwd_walk >= 0) walk_remember(op, n_scalars);       // the choice has just been made
            }
        } else if (op->fwd_walk < K_FWD_CANDIDATES) {
            if (!stream_is_capturing(c.stream) && recheck_should_time(op->fwd_tune, fwd_candidates_of(op), &op->fwd_walk)) {
                walk_remember(op, n_scalars);                              // (the re-check may have rotated another candidate in)
                sh = fwd_candidate_shape(op->fwd_walk, n_scalars / NS);
                const bool ok = recheck_begin(op->fwd_tune, c.stream);
                const int st = launch_tall_fwd_shape<S, E, NS>(op, d, m, n_scalars, sh);
                recheck_end(op->fwd_tune, c.stream, ok && st == JH_OK);
                return st;
            }
            sh = fwd_candidate_shape(op->fwd_walk, n_scalars / NS);
        }
        if (slot >= 0) {                                                   // a timed trial: the caller's own launch between two events
            const bool ok = lazy_begin(op->fwd_tune, slot, c.stream);
            const int st = launch_tall_fwd_shape<S, E, NS>(op, d, m, n_scalars, sh);
            lazy_end(op->fwd_tune, slot, c.stream, ok && st == JH_OK);
            return st;
        }
    }
    return launch_tall_fwd_shape<S, E, NS>(op, d, m, n_scalars, sh);
}

template <typename S, int E, int NS>
int launch_tall_fwd_shape(const jh_blockop *op, void *d, const void *m, int64_t n_scalars, const TallShape &sh)
{
    const bool nt = jh_stream_nt(2.0 * (double)op->nrow * (double)n_scalars * sizeof(S));     // coefficients read + range vector written
    if (sh.wg == 256) return nt ? launch_tall_fwd_u<S, E, NS, true, 256>(op, d, m, n_scalars, sh) : launch_tall_fwd_u<S, E, NS, false, 256>(op, d, m, n_scalars, sh);
    if (sh.wg == 512) return nt ? launch_tall_fwd_u<S, E, NS, true, 512>(op, d, m, n_scalars, sh) : launch_tall_fwd_u<S, E, NS, false, 512>(op, d, m, n_scalars, sh);
    return nt ? launch_tall_fwd_u<S, E, NS, true, 1024>(op, d, m, n_scalars, sh) : launch_tall_fwd_u<S, E, NS, false, 1024>(op, d, m, n_scalars, sh);
}
template <typename S, int E, int NS, int MODE>
int launch_tall_adj(const jh_blockop *op, void *out, const void *in, int64_t n_scalars, int64_t s_begin = 0, int64_t s_end = -1)
{
    if (s_end < 0) s_end = n_scalars;
    if (s_end <= s_begin) return JH_OK;
    const TallShape sh = pick_adj_shape(n_scalars / NS, op->nrow, MODE);
    // coefficients + the range vector; the fused A'A reads the coefficients alone, but at 256 MiB of them nontemporal loads are already 3 % ahead
    // (profiles/exp_r05_nt_small.txt), so it counts them twice as well
    const bool nt = jh_stream_nt(2.0 * (double)op->nrow * (double)n_scalars * sizeof(S));
    if (sh.wg == 256) return nt ? launch_tall_adj_u<S, E, NS, true, MODE, 256>(op, out, in, n_scalars, sh, s_begin, s_end) : launch_tall_adj_u<S, E, NS, false, MODE, 256>(op, out, in, n_scalars, sh, s_begin, s_end);
    if (sh.wg == 512) return nt ? launch_tall_adj_u<S, E, NS, true, MODE, 512>(op, out, in, n_scalars, sh, s_begin, s_end) : launch_tall_adj_u<S, E, NS, false, MODE, 512>(op, out, in, n_scalars, sh, s_begin, s_end);
    return nt ? launch_tall_adj_u<S, E, NS, true, MODE, 1024>(op, out, in, n_scalars, sh, s_begin, s_end) : launch_tall_adj_u<S, E, NS, false, MODE, 1024>(op, out, in, n_scalars, sh, s_begin, s_end);
}

// Does the plain tall adjoint (MODE 0) take the split walk for this operator?  If so, reserve scratch for its slabs PLUS one
// domain-sized temporary behind them and return that temporary: the kernels that have no split variant of their own
// (fused adjoint update, JetSum adjoint) then run "split adjoint into the temporary + a small epilogue" instead of crawling.
template <typename S, int NS>
int split_adjoint_tmp(const jh_blockop *op, int64_t n_scalars, bool c32, void **tmp)
{
    *tmp = nullptr;
    if (op->nrow == 1) return JH_OK;
    const TallShape sh = pick_adj_shape(n_scalars / NS, op->nrow, 0);
    const int64_t gx0 = (n_scalars + (int64_t)sh.unroll * sh.wg * NS - 1) / ((int64_t)sh.unroll * sh.wg * NS);
    int64_t parts = pick_adj_parts(gx0, op->nrow);
    if (parts <= 1) return JH_OK;
    // The caller sends operators off the 16-byte grid to tall_adj(..., mixed = true), whose MODE-0 shapes (launch_tall_adj_mixed: 256 x 1 for rows
    // below 2048 packs, else -- and always for ComplexF32 -- 512 x 2) launch fewer workgroups per part than the all-DIAG shape above and so cut the rows
    // into MORE parts.  Reserve for whichever route asks for most (round-5 advisor finding: 1024 rows of 40001 Float32 reserved 52 slabs, the mixed
    // launch used 64 -- slabs 52..63 lay over the temporary, or the launcher regrew the scratch and freed the memory `tmp` points into).
    {
        const int64_t packs = (n_scalars + NS - 1) / NS;
        const int64_t per_wg = (packs < 2048 && !c32) ? (int64_t)256 * NS : (int64_t)1024 * NS;
        const int64_t parts_mixed = pick_adj_parts((n_scalars + per_wg - 1) / per_wg, op->nrow);
        if (parts_mixed > parts) parts = parts_mixed;
    }
    const int64_t rows_per_part = (op->nrow + parts - 1) / parts;
    parts = (op->nrow + rows_per_part - 1) / rows_per_part;
    const size_t slab_bytes = ((size_t)parts * (size_t)n_scalars * sizeof(S) + 255) / 256 * 256;
    void *base = nullptr;
    JH_TRY(jh_ensure_scratch(slab_bytes + (size_t)n_scalars * sizeof(S), &base));
    *tmp = (char *)base + slab_bytes;
    return JH_OK;
}

// fast path usable?  (tall, all DIAG, uniform rows, 16-byte aligned everything, no conj flags on complex)
}  // namespace
namespace jhb {
bool tall_fast_ok(const jh_blockop *op, const void *rng_ptr, const void *dom_ptr)
{
    if (!(op->tall && op->all_diag && op->uniform_rows)) return false;
    const size_t es = jh_dtype_size(op->dtype);
    const int64_t n = op->row_len[0];
    if (n == 0 || (n * (int64_t)es) % 16 != 0) return false;
    if ((((uintptr_t)rng_ptr) | ((uintptr_t)dom_ptr)) & 15u) return false;
    return op->coeff_aligned16;
}
}  // namespace jhb
namespace {

// mixed tall path usable?  Tall with >= 2 equal rows of ANY elementwise kind (ZERO / IDENTITY / SCALE / DIAG, adjointed or
// not / the Jacobian of SQUARE), everything 16-byte aligned -- the rows a pure-diagonal operator gains when a regularisation
// row (identity, scalar) or a muted shot (zero block) joins it.  Such operators keep the tall kernels' tiling, the fused A'A
// and the one-pass LSQR step instead of dropping to the general M x K kernels.
}  // namespace
namespace jhb {
bool tall_mixed_ok(const jh_blockop *op, const void *rng_ptr, const void *dom_ptr)
{
    if (!(op->tall && op->uniform_rows && op->elementwise) || op->nrow < 2 || op->all_diag) return false;
    const size_t es = jh_dtype_size(op->dtype);
    const int64_t n = op->row_len[0];
    if (n == 0 || (n * (int64_t)es) % 16 != 0) return false;
    if ((((uintptr_t)rng_ptr) | ((uintptr_t)dom_ptr)) & 15u) return false;
    return op->coeff_aligned16;
}
}  // namespace jhb
namespace {

// Rows that are NOT whole 16-byte packs, or do not start on 16-byte boundaries (round 5, session 3): a tall operator of >= 2 equal elementwise rows whose
// block length is odd -- 101^3 Float32 elements -- has three rows in four start off a 16-byte boundary inside the range vector's slab (and, when the
// diagonals are blocks of one slab too, inside the coefficients).  Such operators used to drop to the general kernels' 4-byte-per-lane forms (256 x 1 of
// 524 289 Float32: forward 2.0, adjoint 3.1 TB/s); they now run the MIXED instantiations of the tall kernels, whose accesses are under-aligned packs and
// whose last pack per row is partial (jh_blockop_common.h: ldu / st_pack) -- the same terms in the same order, so the same bits.  All-diagonal operators
// included: the per-row kind switch costs them a scalar branch per row.  The fused forward update, the one-pass step (plain walk), the ranged calls, the
// fused sums and adjoint update and the solver loops (host-driven, graph-replayed, partitioned) take such operators as well (jh_blockop_tall_step_ok).
}  // namespace
namespace jhb {
bool tall_unaligned_ok(const jh_blockop *op, const void *rng_ptr, const void *dom_ptr)
{
    if (!(op->tall && op->uniform_rows && op->elementwise) || op->nrow < 2) return false;
    if (jh_ctx().tall_unaligned == 0) return false;                       // knob: 0 sends such operators to the general kernels as before
    const size_t es = jh_dtype_size(op->dtype);
    const int64_t n = op->row_len[0];
    if (n * (int64_t)es < 16) return false;                               // at least one pack per row
    const size_t sa = jh_dtype_complex(op->dtype) ? es / 2 : es;          // the scalar's alignment
    if ((((uintptr_t)rng_ptr) | ((uintptr_t)dom_ptr)) & (sa - 1)) return false;
    return op->coeff_scalar_aligned;
}
}  // namespace jhb
namespace {

// one shape per kernel for the mixed rows (they are the exception; the all-DIAG instantiations keep their tuned shapes)
template <typename S, int E, int NS>
int launch_tall_fwd_mixed(const jh_blockop *op, void *d, const void *m, int64_t n_scalars, int fmode = 0)
{
    jh_context &c = jh_ctx();
    // late round 4: one pack per lane, two rows per workgroup, COLUMN bands of 32 tiles (128 KiB of each row, then the same tiles of the next
    // row group: k_tall_diag_fwd's ctiles decode) -- against round 3's 256 x 4 packs x 4 rows in a sequential sweep: 1024 x 64^3 5.0 -> 6.1 TB/s,
    // 4096 x 64^3 5.3 -> 6.3, 16384 x 32^3 5.5 -> 6.4, 64 x 128^3 5.6 -> 6.4, 2048 x 128^3 5.6 -> 6.35, 256 x 256^3 5.8 -> 6.4
    // (profiles/exp_r04_mixed_fwd.txt).  Knobs fwd_group / fwd_ctiles override rows per workgroup / tiles per band (0: sequential sweep).
    constexpr int BLK = 256, U = 1;
    // the rows the launch covers: all of them, or (late round 5) the operator's non-zero rows when an eighth or more are zero blocks (knob general_list = 0: all)
    const bool listed = !fmode && c.general_list != 0 && op->dev_rows_nz && op->n_rows_nz * 8 <= op->nrow * 7;   // (f! writes every row)
    const int64_t nrows = listed ? op->n_rows_nz : op->nrow;
    if (nrows == 0) return JH_OK;                                          // every row a zero block: d stays as found (1022)
    // (rows of a few KiB -- traces rather than volumes --: more rows per workgroup, so that a workgroup still moves ~16 KiB: 262144 x 513 Float32 4.3 -> see
    // profiles/bench_unaligned_r05.txt)
    const int64_t row_bytes = n_scalars * (int64_t)sizeof(S);
    int64_t G = c.fwd_group > 0 ? c.fwd_group : (row_bytes <= 2560 ? 8 : (row_bytes <= 5120 ? 4 : 2));
    if (G > nrows) G = nrows;
    const int64_t gx = (n_scalars + (int64_t)U * BLK * NS - 1) / ((int64_t)U * BLK * NS);
    int64_t gy = (nrows + G - 1) / G;
    while (gx * gy * BLK >= ((int64_t)1 << 32) && G < nrows) { G *= 2; gy = (nrows + G - 1) / G; }
    JH_REQUIRE(gx * gy * BLK < (int64_t)1 << 32, "tall forward: grid of %lld workgroups is too large", (long long)(gx * gy));
    int64_t ctiles = c.fwd_ctiles >= 0 ? c.fwd_ctiles : 32;
    if (ctiles > gx) ctiles = gx;
    // Round 6: rows that are not whole packs (an odd block length) on lanes anchored to each row's own 16-byte grid (k_tall_fwd_anchored): from rows of
    // 64 KiB on (below, a row's two partial slots are too large a share), not for the listed walk of mostly-zero operators.  Knob fwd_anchor: -1 this rule, 0 never, 1 always.
    if (!listed && n_scalars % NS != 0 && n_scalars >= 2 * NS && (((uintptr_t)m) & 15u) == 0 &&
        (c.fwd_anchor > 0 || (c.fwd_anchor < 0 && row_bytes >= (64 << 10)))) {
        const int64_t Ga = c.fwd_group > 0 ? c.fwd_group : 1;                            // (rows of one class are NS rows apart: one row per workgroup streams best -- profiles/exp_r06_fwd_anchor.txt)
        const int64_t slots = (n_scalars + 2 * NS - 2) / NS;                             // a row spans at most this many 16-byte slots
        const int64_t ax = (slots + BLK - 1) / BLK;
        const int64_t ay = NS * ((((op->nrow + NS - 1) / NS) + Ga - 1) / Ga);            // groups: NS phase classes x chunks of Ga rows of a class
        int64_t act = c.fwd_ctiles >= 0 ? c.fwd_ctiles : (row_bytes >= ((int64_t)32 << 20) ? 32 : 128);
        if (act > ax) act = ax;
        if (act < 1) act = 1;
        if (ax * ay * BLK < ((int64_t)1 << 32)) {
            const int base_phase = (int)((((uintptr_t)d) / sizeof(S)) % NS);
            c.last_fwd_walk = 3;
            c.last_fwd_rows_per_wg = Ga;
            // diagonals in ONE slab laid out like the range vector share its phase row by row: their loads are aligned as well and stream (nontemporal);
            // separately allocated diagonals are read under-aligned -- temporal, so that a line two waves share is still in L2 (launch_tall_fwd_mixed)
            const bool coef_like_d = op->all_diag && op->diag_strided && op->diag_stride_elems * E == n_scalars &&
                                     (int)((((uintptr_t)op->blocks[0].coeff) / sizeof(S)) % NS) == base_phase;
            const bool nt_loads = c.ua_nt == 1 || (c.ua_nt < 0 && coef_like_d && jh_stream_nt(2.0 * (double)op->nrow * (double)row_bytes));
            if (!nt_loads)
                hipLaunchKernelGGL((k_tall_fwd_anchored<S, E, NS, false, BLK>), dim3((unsigned)(ax * ay)), dim3(BLK), 0, c.stream, op->dev_blocks, op->nrow, (int)Ga,
                                   (const S *)m, (S *)d, n_scalars, (unsigned)ax, (unsigned)ay, (unsigned)act, base_phase, fmode);
            else
                hipLaunchKernelGGL((k_tall_fwd_anchored<S, E, NS, true, BLK>), dim3((unsigned)(ax * ay)), dim3(BLK), 0, c.stream, op->dev_blocks, op->nrow, (int)Ga,
                                   (const S *)m, (S *)d, n_scalars, (unsigned)ax, (unsigned)ay, (unsigned)act, base_phase, fmode);
            JH_CHECK_HIP(hipGetLastError());
            return JH_OK;
        }
    }
    c.last_fwd_walk = ctiles ? 2 : 0;
    c.last_fwd_rows_per_wg = G;
#define JH_FWD_MIXED(NTV)                                                                                                                   \
    hipLaunchKernelGGL((k_tall_diag_fwd<S, E, NS, U, NTV, BLK, true>), dim3((unsigned)(gx * gy)), dim3(BLK), 0, c.stream, op->dev_blocks,        \
                       nrows, (int)G, listed ? reinterpret_cast<const S *>(op->dev_rows_nz) : (const S *)nullptr, (int64_t)(listed ? -1 : 0), (const S *)m, (S *)d, \
                       n_scalars, (unsigned)gx, (unsigned)gy, 1u, (unsigned)ctiles, fmode)
    // Rows off the 16-byte grid take TEMPORAL LOADS: a 128-byte line that two neighbouring waves share is then still in L2 when the second one asks --
    // streamed, it came from HBM twice (forward 256 x 255^3 5.82 -> 5.92 TB/s, 512 x 127^3 5.02 -> 5.11, 1024 x 101^3 5.30 -> 5.47; aligned rows: no
    // difference either way; profiles/bench_unaligned_r05.txt).  The STORES stay streaming stores (k_tall_diag_fwd).  Knob ua_nt: -1 this rule,
    // 0 / 1 temporal / nontemporal loads always.
    const bool off_grid = (n_scalars * (int64_t)sizeof(S)) % 16 != 0 || !op->coeff_aligned16 || ((((uintptr_t)d) | ((uintptr_t)m)) & 15u) != 0;
    if (c.ua_nt == 0 || (c.ua_nt < 0 && off_grid)) JH_FWD_MIXED(false); else JH_FWD_MIXED(true);
#undef JH_FWD_MIXED
    JH_CHECK_HIP(hipGetLastError());
    return JH_OK;
}

template <typename S, int E, int NS, int MODE, int BLK, int U, int DEPTH, bool TAIL = true>
int launch_tall_adj_mixed_u(const jh_blockop *op, void *out, const void *in, int64_t n_scalars, int64_t s_begin, int64_t s_end)
{
    jh_context &c = jh_ctx();
    const int64_t gx = (s_end - s_begin + (int64_t)U * BLK * NS - 1) / ((int64_t)U * BLK * NS);
    const int from_found = (MODE == 0) ? c.adj_from_found : 0;                        // continue from what `out` holds (a wide operator's forward)
    int64_t parts = (from_found || s_end - s_begin < NS) ? 1 : pick_adj_parts(gx, op->nrow), rows_per_part = 0;   // many rows of small blocks: split-row walk (not for a range
                                                                                                                  // shorter than one pack: it loads from before its begin)
    if (MODE == 0 && BLK == 256 && c.adj_thin_mixed && parts == 1 && c.adj_split < 0 && !from_found && s_end - s_begin >= NS && op->nrow >= 256 &&
        gx >= c.cu_count && gx < 2 * (int64_t)c.cu_count)
        parts = 2;                                                                                                // (the thin shape at one workgroup per CU: latency-bound in one part)
    const int64_t part_stride = s_end - s_begin;
    void *slabs = nullptr;
    if (parts > 1) {
        rows_per_part = (op->nrow + parts - 1) / parts;
        parts = (op->nrow + rows_per_part - 1) / rows_per_part;
        JH_TRY(adj_slabs(out, (size_t)parts * (size_t)part_stride * sizeof(S), &slabs));
    }
    c.last_adj_parts = parts;
    c.last_adj_launches = 1;
#define JH_ADJ_MIXED(NTV)                                                                                                                   \
    hipLaunchKernelGGL((k_tall_diag_adj<S, E, NS, U, DEPTH, NTV, MODE, BLK, true, TAIL>), dim3((unsigned)gx, (unsigned)parts), dim3(BLK), 0, c.stream, \
                       op->dev_blocks, op->nrow, (const S *)nullptr, (int64_t)0, (S *)out, (const S *)in, n_scalars, 0, s_begin, s_end,            \
                       (int64_t)0, op->nrow, from_found, rows_per_part, (S *)slabs, part_stride, (S)(MODE == 0 ? c.adj_in_scale : 1.0))
    // (rows off the 16-byte grid: temporal loads, see launch_tall_fwd_mixed -- from 32 MiB rows on: 256 x 255^3 adjoint 5.83 -> 6.17 TB/s, but 512 x 127^3
    // 5.97 -> 5.73 and 1024 x 101^3 5.70 -> 5.67 on the same boxes)
    const bool off_grid = (n_scalars * (int64_t)sizeof(S)) % 16 != 0 || !op->coeff_aligned16 || ((((uintptr_t)out) | ((uintptr_t)in)) & 15u) != 0;
    // (the shape without the partial-pack logic only ever sees aligned rows: no temporal instantiation of it)
    if constexpr (TAIL) {
        if (c.ua_nt == 0 || (c.ua_nt < 0 && off_grid && n_scalars * (int64_t)sizeof(S) >= ((int64_t)32 << 20))) JH_ADJ_MIXED(false); else JH_ADJ_MIXED(true);
    } else {
        JH_ADJ_MIXED(true);
    }
#undef JH_ADJ_MIXED
    JH_CHECK_HIP(hipGetLastError());
    if (parts > 1) return launch_fold_parts<S, NS>(slabs, part_stride, parts, out, s_begin, s_end);
    return JH_OK;
}

template <typename S, int E, int NS, int MODE>
int launch_tall_adj_mixed(const jh_blockop *op, void *out, const void *in, int64_t n_scalars, int64_t s_begin = 0, int64_t s_end = -1)
{
    if (s_end < 0) s_end = n_scalars;
    if (s_end <= s_begin) return JH_OK;
    // the fused normal operator reads ONE stream: fat workgroups with more rows in flight once the blocks are big (like the
    // all-DIAG shapes of pick_adj_shape); everything else 512 x 2 x 2
    // (ComplexF32 with its per-row kind switch: two rows in flight, four spilled 20 bytes per lane)
    if constexpr (MODE == 1) {
        constexpr int DEPTH = (E == 2 && sizeof(S) == 4) ? 2 : 4;
        if (n_scalars / NS >= ((int64_t)1 << 22)) {
            // (four rows in flight only on whole, aligned packs: k_tall_diag_adj's TAIL)
            const bool off_grid = (n_scalars * (int64_t)sizeof(S)) % 16 != 0 || !op->coeff_aligned16 || ((((uintptr_t)out) | ((uintptr_t)in)) & 15u) != 0 ||
                                  (s_end - s_begin) % NS != 0;
            if (off_grid) return launch_tall_adj_mixed_u<S, E, NS, MODE, 1024, 4, 2>(op, out, in, n_scalars, s_begin, s_end);
            return launch_tall_adj_mixed_u<S, E, NS, MODE, 1024, 4, DEPTH, DEPTH == 2>(op, out, in, n_scalars, s_begin, s_end);
        }
    }
    // rows of a few KiB (traces rather than volumes: 1001 or 2049 samples): thin workgroups, one pack per lane and four rows in flight -- the 512 x 2 shape issued
    // its (clamped) loads for 1024 packs where a row has 129: 262144 x 513 Float32 adjoint 2.1 -> see profiles/bench_unaligned_r05.txt (the split-row walk
    // supplies the workgroups)
    // (round 6, tried and dropped: the thin shape whenever 512 x 2 launches fewer than two workgroups per CU -- 256 rows of 2 MiB with an identity row in four: the
    // ordered walk on 512 thin workgroups ran the fused A'A at 4.15 TB/s where the split walk over sixteen row parts, which 512 x 2 falls into there, runs 4.6;
    // profiles/rocprof_r06_thin_summary.md)
    // (round 6, kept for the ADJOINT only: rows of 1-8 MiB, where 512 x 2 launches fewer workgroups than the chip has CUs and falls into the split walk although
    // thin workgroups fill it in one ordered part -- 256 x 2 MiB with an identity row: adjoint 5.42 (16 parts) -> 6.6 TB/s, what the chain kernel of the same
    // shape showed, tools/exp_chain_vs_mixed.py; two parts when that leaves under two workgroups per CU, as for the chains)
    const int64_t packs_here = (s_end - s_begin + NS - 1) / NS;
    const bool thin_fills = MODE == 0 && jh_ctx().adj_thin_mixed && packs_here / 2048 < jh_ctx().cu_count && packs_here / 256 >= jh_ctx().cu_count;
    if ((packs_here < 2048 || thin_fills) && !(E == 2 && sizeof(S) == 4))
        return launch_tall_adj_mixed_u<S, E, NS, MODE, 256, 1, 4>(op, out, in, n_scalars, s_begin, s_end);
    return launch_tall_adj_mixed_u<S, E, NS, MODE, 512, 2, 2>(op, out, in, n_scalars, s_begin, s_end);
}


}  // namespace

namespace jhb {

#define JH_BY_DTYPE(CALL)                                                  \
    switch (op->dtype) {                                                   \
    case JH_F32: return CALL(float, 1, 4);                                 \
    case JH_F64: return CALL(double, 1, 2);                                \
    case JH_C32: return CALL(float, 2, 4);                                 \
    case JH_C64: return CALL(double, 2, 2);                                \
    }                                                                      \
    return jh_fail(JH_ERR_INVALID, "unknown dtype %d", op->dtype)

int tall_fwd(const jh_blockop *op, void *d, const void *m)
{
    const int64_t n = op->row_len[0];
#define JH_CALL(S, E, NS) launch_tall_fwd<S, E, NS>(op, d, m, n * E)
    JH_BY_DTYPE(JH_CALL);
#undef JH_CALL
}

int tall_fwd_mixed(const jh_blockop *op, void *d, const void *m, int fmode)
{
    const int64_t n = op->row_len[0];
#define JH_CALL(S, E, NS) launch_tall_fwd_mixed<S, E, NS>(op, d, m, n * E, fmode)
    JH_BY_DTYPE(JH_CALL);
#undef JH_CALL
}

int tall_adj(const jh_blockop *op, void *out, const void *in, int mode, bool mixed, int64_t first_elem, int64_t end_elem)
{
    const int64_t n = op->row_len[0];
    if (mixed && first_elem == 0 && (end_elem < 0 || end_elem == n)) {                     // rows of up to 4 MiB: the chain kernels' packed row records (jh_tall_chain.hip)
        bool took = false;
        JH_TRY(bare_chain(op, out, in, mode, &took));
        if (took) return JH_OK;
    }
#define JH_CALL(S, E, NS)                                                                                                                    \
    (mode == 0 ? (mixed ? launch_tall_adj_mixed<S, E, NS, 0>(op, out, in, n * E, first_elem * E, end_elem < 0 ? -1 : end_elem * E)           \
                        : launch_tall_adj<S, E, NS, 0>(op, out, in, n * E, first_elem * E, end_elem < 0 ? -1 : end_elem * E))                 \
               : (mixed ? launch_tall_adj_mixed<S, E, NS, 1>(op, out, in, n * E, first_elem * E, end_elem < 0 ? -1 : end_elem * E)           \
                        : launch_tall_adj<S, E, NS, 1>(op, out, in, n * E, first_elem * E, end_elem < 0 ? -1 : end_elem * E)))
    JH_BY_DTYPE(JH_CALL);
#undef JH_CALL
}

int fold_parts(int dtype, const void *parts, int64_t part_stride, int64_t nparts, void *out, int64_t s_begin, int64_t s_end)
{
    if (dtype == JH_F32 || dtype == JH_C32) return launch_fold_parts<float, 4>(parts, part_stride, nparts, out, s_begin, s_end);
    return launch_fold_parts<double, 2>(parts, part_stride, nparts, out, s_begin, s_end);
}

int split_adjoint_tmp(const jh_blockop *op, void **tmp)
{
    const int64_t ns = op->row_len[0] * (jh_dtype_complex(op->dtype) ? 2 : 1);
    if (op->dtype == JH_F32 || op->dtype == JH_C32) return ::split_adjoint_tmp<float, 4>(op, ns, op->dtype == JH_C32, tmp);
    return ::split_adjoint_tmp<double, 2>(op, ns, false, tmp);
}
#undef JH_BY_DTYPE

}  // namespace jhb

bool jh_blockop_tall_fast(const jh_blockop *op, const void *rng_ptr, const void *dom_ptr)
{
    return tall_fast_ok(op, rng_ptr, dom_ptr) || (tall_mixed_ok(op, rng_ptr, dom_ptr) && !(op->nonlinear && !op->pointed));
}

// what the fused passes accept (forward update, one-pass step -- whole-vector or ranged --, and the solver loops built on them): the above, or rows off the
// 16-byte pack grid on the under-aligned MIXED instantiations (tall_unaligned_ok)
bool jh_blockop_tall_step_ok(const jh_blockop *op, const void *rng_ptr, const void *dom_ptr)
{
    return jh_blockop_tall_fast(op, rng_ptr, dom_ptr) || (tall_unaligned_ok(op, rng_ptr, dom_ptr) && !(op->nonlinear && !op->pointed));
}

// all-DIAG tall operators only (the caller checks: jh_lsqr.hip, cg_graph_impl); one pack per lane, 8 rows in flight
int jh_launch_cg_normal(const jh_blockop *op, jh_bvec *p, const jh_bvec *s, jh_bvec *y, const jh_cg_dev *st, double *partials, int64_t *nparts)
{
    jh_context &c = jh_ctx();
    JH_REQUIRE(op->all_diag && (tall_fast_ok(op, nullptr, p->data) || tall_unaligned_ok(op, nullptr, p->data)), "cg normal pass: needs a tall all-DIAG operator with equal blocks");
    const int64_t n = op->row_len[0];
    const int64_t packs = (n * (int64_t)jh_dtype_size(op->dtype) + 15) / 16;
    // launch-bound domains: thin workgroups with eight rows in flight; from 2 MiB blocks on the fused normal operator's shapes
    const int shape = packs >= ((int64_t)1 << 22) ? 2 : (packs >= ((int64_t)1 << 17) ? 1 : 0);
    const int64_t per_wg = shape == 2 ? 4096 : (shape == 1 ? 1024 : 256);
    const int64_t grid = (packs + per_wg - 1) / per_wg;
    JH_REQUIRE(grid >= 1 && grid < ((int64_t)1 << 22), "cg normal pass: domain of %lld elements is out of range", (long long)n);
    *nparts = grid;
#define JH_CGN_S(S, E, NS, DEPTH, BLK, U, NTV)                                                                                                  \
    hipLaunchKernelGGL((k_cg_normal<S, E, NS, DEPTH, BLK, U, NTV>), dim3((unsigned)grid), dim3(BLK), 0, c.stream, op->dev_blocks, op->nrow,      \
                       op->diag_strided ? (const S *)op->blocks[0].coeff : (const S *)nullptr, op->diag_stride_elems * E, (S *)p->data,          \
                       (const S *)s->data, (S *)y->data, n * E, st, partials)
    // the coefficients of an operator that an iteration re-reads and that fit the Infinity Cache are loaded TEMPORAL (jh_stream_nt)
    const bool nt = jh_stream_nt(2.0 * (double)op->nrow * (double)n * (double)jh_dtype_size(op->dtype));
#define JH_CGN(S, E, NS)                                                                                                                        \
    do {                                                                                                                                        \
        if (shape == 2) JH_CGN_S(S, E, NS, ((E == 2 && sizeof(S) == 4) ? 2 : 4), 1024, 4, true);                                                \
        else if (shape == 1) { if (nt) JH_CGN_S(S, E, NS, 2, 512, 2, true); else JH_CGN_S(S, E, NS, 2, 512, 2, false); }                        \
        else { if (nt) JH_CGN_S(S, E, NS, 8, 256, 1, true); else JH_CGN_S(S, E, NS, 8, 256, 1, false); }                                        \
    } while (0)
    switch (op->dtype) {
    case JH_F32: JH_CGN(float, 1, 4); break;
    case JH_F64: JH_CGN(double, 1, 2); break;
    case JH_C32: JH_CGN(float, 2, 4); break;
    case JH_C64: JH_CGN(double, 2, 2); break;
    default: return jh_fail(JH_ERR_INVALID, "cg normal pass: unknown dtype %d", op->dtype);
    }
#undef JH_CGN_S
#undef JH_CGN
    JH_CHECK_HIP(hipGetLastError());
    return JH_OK;
}

extern "C" {

int jh_blockop_tune_get(const jh_blockop *op, const char *name, int64_t *value)
{
    JH_REQUIRE(op && name && value, "jh_blockop_tune_get: null argument");
    if (!strcmp(name, "fwd_walk")) *value = op->fwd_walk;                       // -1: not chosen yet
    else if (!strcmp(name, "fwd_walk_inherited")) *value = op->walk_inherited ? 1 : 0;   // the choice came from an earlier operator of the same shape
    else if (!strcmp(name, "fwd_trials")) *value = op->fwd_tune.launched;
    else if (!strcmp(name, "fwd_switches")) *value = op->fwd_tune.switches;       // times the periodic re-check rotated another walk in
    else if (!strcmp(name, "fwd_playoff")) *value = op->fwd_tune.playoff[0] >= 0 ? op->fwd_tune.playoff[0] * 16 + op->fwd_tune.playoff[1] : -1;
    else if (!strcmp(name, "step_trials")) *value = op->step_tune.launched;
    else if (!strcmp(name, "upd_walk")) *value = op->upd_walk;
    else if (!strcmp(name, "step_mode")) *value = op->step_mode;
    else if (!strcmp(name, "gen_walk_fwd")) *value = op->gen_walk[0];            // sparse grids: -1 not chosen, 0 four-line step lists, 1 per-line lists, 2 plain walk
    else if (!strcmp(name, "gen_walk_adj")) *value = op->gen_walk[1];
    else if (!strcmp(name, "gen_trials")) *value = op->gen_tune[0].launched + op->gen_tune[1].launched;
    else return jh_fail(JH_ERR_INVALID, "jh_blockop_tune_get: unknown per-operator knob '%s'", name);
    return JH_OK;
}

int jh_blockop_tune_set(jh_blockop *op, const char *name, int64_t value)
{
    JH_REQUIRE(op && name, "jh_blockop_tune_set: null argument");
    if (!strcmp(name, "fwd_walk")) {
        JH_REQUIRE(value >= -1 && value < K_FWD_CANDIDATES, "jh_blockop_tune_set: fwd_walk must be -1 (measure again) or 0..%d", K_FWD_CANDIDATES - 1);
        lazy_reset(op->fwd_tune);
        op->fwd_walk = (int)value;
        op->walk_measure_again = value < 0;                                 // -1: THIS operator measures, whatever operators of its shape found before
        op->walk_inherited = false;
    } else if (!strcmp(name, "upd_walk")) {
        JH_REQUIRE(value >= -1 && value <= 1, "jh_blockop_tune_set: upd_walk must be -1, 0 or 1");
        op->upd_walk = (int)value;
        op->upd_trials = value < 0 ? 0 : 2;
    } else if (!strcmp(name, "step_mode")) {
        JH_REQUIRE(value >= -1 && value <= 2, "jh_blockop_tune_set: step_mode must be -1 (measure), 0 (plain walk), 1 (XCD-contiguous tiles) or 2 (chained row chunks)");
        lazy_reset(op->step_tune);
        op->step_span = 0;
        op->step_mode = (int)value;
    } else if (!strcmp(name, "gen_walk_fwd") || !strcmp(name, "gen_walk_adj")) {
        JH_REQUIRE(value >= -1 && value <= 2, "jh_blockop_tune_set: %s must be -1 (measure), 0 (four-line step lists), 1 (per-line lists) or 2 (plain walk)", name);
        const int dir = !strcmp(name, "gen_walk_adj") ? 1 : 0;
        lazy_reset(op->gen_tune[dir]);
        op->gen_walk[dir] = (int)value;
    } else return jh_fail(JH_ERR_INVALID, "jh_blockop_tune_set: unknown per-operator knob '%s'", name);
    return JH_OK;
}


}  // extern "C"
