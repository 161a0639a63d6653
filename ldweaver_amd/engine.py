"""Thin object wrapper over the C ABI: one Engine = one ldw_ctx on one GPU."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L


def _is_torch(x):
    return hasattr(x, "data_ptr") and hasattr(x, "is_cuda")


class Engine:
    def __init__(self, device: int = 0, stream=None):
        self._ctx = C.c_void_p()
        L.check(L.lib().ldw_ctx_create(int(device), C.byref(self._ctx)))
        self.device = int(device)
        self.L = self.N = 0
        if stream is not None:
            self.set_stream(stream)

    # -- lifetime ------------------------------------------------------------
    def close(self, trim: bool = False):
        """Destroy the context.  Its large device blocks go to the process-wide free list for the next engine (capped; trimmed to LDW_DEVPOOL_IDLE_GB when the
        process's last context goes); trim=True gives everything back to the runtime at once — for a process that shares the GPU with other allocators
        (torch, RCCL buffers, other ranks on the same device)."""
        if self._ctx:
            L.lib().ldw_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p()
            if trim:
                n = C.c_int64(0)
                L.lib().ldw_host_trim(None, C.byref(n))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_stream(self, stream):
        """stream: a raw hipStream_t value (int), e.g. torch.cuda.current_stream().cuda_stream; None/0 = own stream."""
        L.check(L.lib().ldw_ctx_set_stream(self._ctx, C.c_void_p(int(stream) if stream else 0)))

    def sync(self):
        L.check(L.lib().ldw_ctx_sync(self._ctx))

    def last_timing(self):
        t = np.zeros(4)
        L.check(L.lib().ldw_ctx_last_timing(self._ctx, L.ptr(t)))
        return dict(gemm_ms=t[0], epilogue_ms=t[1], select_ms=t[2], total_ms=t[3])

    def gemm_stats(self, reset: bool = False):
        """Launches and EXECUTED int8 operations of the block-wide GEMMs since the last reset (ldw_gemm_stats)."""
        v = np.zeros(6)
        L.check(L.lib().ldw_gemm_stats(self._ctx, L.ptr(v), int(reset)))
        return dict(apx_launches=int(v[0]), apx_ops=float(v[1]), bits_launches=int(v[2]), bits_ops=float(v[3]), band_launches=int(v[4]),
                    apx_table_launches=int(v[5]))

    def hamming_stats(self):
        """What the last hamming_weights call did (ldw_hamming_stats): columns, stage times (HIP events), algorithmic bytes around the GEMM, host wall ms."""
        v = np.zeros(8)
        L.check(L.lib().ldw_hamming_stats(self._ctx, L.ptr(v)))
        return dict(columns=int(v[0]), k_padded=int(v[1]), pre_ms=float(v[2]), gemm_ms=float(v[3]), post_ms=float(v[4]), pre_bytes=float(v[5]), post_bytes=float(v[6]),
                    wall_ms=float(v[7]))

    def counters(self):
        v = np.zeros(8, dtype=np.int64)
        L.check(L.lib().ldw_ctx_counters2(self._ctx, L.ptr(v)))
        return dict(spec_misses=int(v[0]), fused_blocks=int(v[1]), unfused_blocks=int(v[2]), screen_violations=int(v[3]),
                    mixed_blocks=int(v[4]), apx_blocks=int(v[5]), apx_units_listed=int(v[6]), apx_pairs_listed=int(v[7]))

    def reset_speculation(self):
        """Forget the bucket guesses / threshold table earlier passes left behind: the next pass runs as a job's first pass."""
        L.check(L.lib().ldw_reset_speculation(self._ctx))

    def path_report(self):
        """Which execution path the blocks took so far and, if the approximate path is off, the gate that failed."""
        v = np.zeros(8, dtype=np.int64)
        buf = C.create_string_buffer(200)
        L.check(L.lib().ldw_path_report(self._ctx, L.ptr(v), buf, 200))
        return dict(apx_blocks=int(v[0]), mixed_blocks=int(v[1]), plain_blocks=int(v[2]), fused_blocks=int(v[3]), spec_misses=int(v[4]),
                    probe_blocks=int(v[5]), pairs_listed=int(v[6]), units_listed=int(v[7]), apx_gate=buf.value.decode())

    def set_prune(self, on: bool):
        """Tile pruning of the approximate path (default on): rows ordered by minor-state weight, rare x rare tiles never computed."""
        L.check(L.lib().ldw_set_prune(self._ctx, int(bool(on))))

    def prune_report(self):
        v = np.zeros(4, dtype=np.int64)
        L.check(L.lib().ldw_prune_report(self._ctx, L.ptr(v)))
        return dict(ordered_blocks=int(v[0]), tiles_pruned=int(v[1]), tiles_total=int(v[2]), on=bool(v[3]))

    def sr_pairs(self, blocks, sr_dist: float, n_rows: int | None = None):
        """(a, b) int32 CUDA tensors: the index columns of the short-range table of a pass over `blocks` (in their order), rebuilt from the
        positions alone (ldw_sr_pairs_fill).  What rank 0 of a multi-GPU run does instead of receiving them."""
        import torch
        bl = L.as_c(blocks, np.int32).reshape(-1, 4)
        n = C.c_int64(0)
        if n_rows is None:
            L.check(L.lib().ldw_sr_pairs_fill(self._ctx, L.ptr(bl), len(bl), float(sr_dist), None, None, 0, C.byref(n)))
            n_rows = int(n.value)
        dev = torch.device("cuda", self.device)
        a = torch.empty(max(1, n_rows), dtype=torch.int32, device=dev)
        b = torch.empty(max(1, n_rows), dtype=torch.int32, device=dev)
        torch.cuda.synchronize(dev)
        L.check(L.lib().ldw_sr_pairs_fill(self._ctx, L.ptr(bl), len(bl), float(sr_dist), C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), n_rows, C.byref(n)))
        assert int(n.value) == n_rows, (int(n.value), n_rows)
        return a[:n_rows], b[:n_rows]

    def set_span(self, on: bool, max_blocks: int = 0, corners: bool = False, diag_split: bool = False):
        """Spans (default on): consecutive long-range-only block pairs of one block row run as one launch sequence; results never depend on it.
        corners: corner block pairs join them too (their short-range pairs through SR sub-passes; slower, off by default).
        diag_split: diagonal blocks run as SR sub-pass + weight-ordered long-range pass (off by default)."""
        L.check(L.lib().ldw_set_span(self._ctx, (1 if on else 0) | (2 if (on and corners) else 0) | (4 if (on and diag_split) else 0), int(max_blocks)))

    def span_report(self):
        v = np.zeros(4, dtype=np.int64)
        L.check(L.lib().ldw_span_report(self._ctx, L.ptr(v)))
        return dict(spans=int(v[0]), blocks=int(v[1]), redone=int(v[2]), on=bool(v[3]))

    def overflow_report(self):
        """Blocks / span segments redone because a device list overflowed: pair lists, the maybe list (worst-case sized: 0 unless
        LDW_MAYBE_CAP is set), and whether the maybe list is switched off for the rest of the pass."""
        v = np.zeros(4, dtype=np.int64)
        L.check(L.lib().ldw_overflow_report(self._ctx, L.ptr(v)))
        return dict(pair_list=int(v[0]), maybe_list=int(v[1]), maybe_off=bool(v[2]), maybe_entries=int(v[3]))

    @staticmethod
    def set_pair_cap(cap: int):
        """Tests only: a fixed capacity of the approximate path's pair lists (0: automatic), process-wide."""
        L.check(L.lib().ldw_set_pair_cap(int(cap)))

    def snp_bounds(self):
        """(L, 2, 2) array [snp, RXY reading (intended, reference), partner kind (2, 3 states)]: the largest MI the SNP can reach."""
        out = np.zeros(4 * self.L)
        L.check(L.lib().ldw_snp_bounds(self._ctx, L.ptr(out), out.size))
        return out.reshape(self.L, 2, 2)

    # -- test hooks: the bounds of the default path as functions (BOUNDS.md; csrc/ldw_debug.hip) --------------------
    APX_PARAM_NAMES = ("F", "e_last", "delta", "lost_units", "total_fixed", "neff", "apx_EG", "apx_dfac", "apx_s1", "apx_c1", "apx_W", "apx_unit", "scr_scale",
                       "scr_shift_exact", "scr_scale_exact", "flags", "lo_abs_sum", "lo_bound", "apx_MU", "nlimbs")

    def debug_apx_params(self, want_weights: bool = True):
        """(params[20], V[N], V'[N]): the constants of the approximate screen's bound for the current weights (names: APX_PARAM_NAMES) and, per sequence, the
        exact fixed-point weight and its dual-digit approximation."""
        out = np.zeros(20)
        V = np.zeros(self.N if want_weights else 0, dtype=np.int64)
        Va = np.zeros(self.N if want_weights else 0, dtype=np.int64)
        L.check(L.lib().ldw_debug_apx_params(self._ctx, L.ptr(out), L.ptr(V) if want_weights else None, L.ptr(Va) if want_weights else None, self.N))
        return out, V, Va

    def debug_rows(self):
        """(row0[L + 1], slot_meta[L]): first indicator row of every SNP; rows | uqe flags << 3 | slot states << 8."""
        row0 = np.zeros(self.L + 1, dtype=np.int32)
        meta = np.zeros(self.L, dtype=np.uint32)
        L.check(L.lib().ldw_debug_rows(self._ctx, L.ptr(row0), L.ptr(meta), self.L + 1))
        return row0, meta

    def debug_apx_gemm(self, rows_t, rows_f):
        """gemm_apx_kernel over the given indicator rows: int32 [len(rows_t), len(rows_f)] sums of the dual-digit weights in units of 2^e_last."""
        rt = np.ascontiguousarray(rows_t, dtype=np.int32)
        rf = np.ascontiguousarray(rows_f, dtype=np.int32)
        out = np.zeros((len(rt), len(rf)), dtype=np.int32)
        L.check(L.lib().ldw_debug_apx_gemm(self._ctx, L.ptr(rt), len(rt), L.ptr(rf), len(rf), L.ptr(out)))
        return out

    def debug_screen_bound(self, kind: int, na: int, nb: int, g, pa, pb, pX, pY, rr, params, masks=None):
        """The engine's screen / evaluation device functions on caller-made joint tables (ldw_debug_screen_bound): kind 0 / 1 the approximate path's upper
        bounds (straight-line / predicated), 2 / 3 the fp32 MI of the exact-limb screens, 4 the fp64 value the engine emits."""
        g = np.ascontiguousarray(g, dtype=np.int64).reshape(-1, 16)
        n = len(g)
        pa = np.ascontiguousarray(pa, dtype=np.int64).reshape(n, 5)
        pb = np.ascontiguousarray(pb, dtype=np.int64).reshape(n, 5)
        pX = np.ascontiguousarray(pX, dtype=np.float32).reshape(n, 5)
        pY = np.ascontiguousarray(pY, dtype=np.float32).reshape(n, 5)
        rr = np.ascontiguousarray(rr, dtype=np.float64).reshape(n, 3)
        params = np.ascontiguousarray(params, dtype=np.float64).reshape(20)
        mk = None if masks is None else np.ascontiguousarray(masks, dtype=np.uint32).reshape(n, 2)
        out = np.zeros(n, dtype=np.float32)
        out64 = np.zeros(n, dtype=np.float64)
        L.check(L.lib().ldw_debug_screen_bound(self._ctx, int(kind), int(na), int(nb), n, L.ptr(g), L.ptr(pa), L.ptr(pb), L.ptr(pX), L.ptr(pY), L.ptr(rr),
                                               L.ptr(mk), L.ptr(params), L.ptr(out), L.ptr(out64)))
        return out64 if kind == 4 else out

    def write_links_tsv(self, which: int, path: str, append: bool = True, nthreads: int = 0):
        """The context's sr (0) / lr (1) table as `pos1 pos2 clust1 clust2 len MI` rows (write.table format); (rows, bytes)."""
        n, nb = C.c_int64(0), C.c_int64(0)
        L.check(L.lib().ldw_write_links_tsv(self._ctx, int(which), str(path).encode(), int(bool(append)), int(nthreads), C.byref(n), C.byref(nb)))
        return int(n.value), int(nb.value)

    def host_trim(self) -> int:
        """Release the host memory the library keeps between calls (writer pool, pinned fetch arena); call between jobs.  Bytes released."""
        n = C.c_int64(0)
        L.check(L.lib().ldw_host_trim(self._ctx, C.byref(n)))
        return int(n.value)

    def lr_stream_begin(self, path: str, append: bool = True, nthreads: int = 0):
        """lr_links.tsv appended while the next ``mi_all_pairs`` runs, item by item (the reference appends per block: R/computePairwiseMI.R:362)."""
        L.check(L.lib().ldw_lr_stream_begin(self._ctx, str(path).encode(), int(bool(append)), int(nthreads)))

    def lr_stream_end(self):
        """Wait for the streaming writer; (rows, bytes, blocks whose rows are in the file)."""
        n, nb, blk = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        L.check(L.lib().ldw_lr_stream_end(self._ctx, C.byref(n), C.byref(nb), C.byref(blk)))
        return int(n.value), int(nb.value), int(blk.value)

    def write_links_tsv_begin(self, which: int, path: str, append: bool = True, nthreads: int = 0):
        """Fetch the table now, derive / format / write it on host threads while the caller goes on (``write_links_tsv_end`` waits)."""
        L.check(L.lib().ldw_write_links_tsv_begin(self._ctx, int(which), str(path).encode(), int(bool(append)), int(nthreads)))

    def write_links_tsv_end(self):
        """(rows, bytes) of the table started by ``write_links_tsv_begin``; raises what its writer reported."""
        n, nb = C.c_int64(0), C.c_int64(0)
        L.check(L.lib().ldw_write_links_tsv_end(self._ctx, C.byref(n), C.byref(nb)))
        return int(n.value), int(nb.value)

    def set_overlap(self, on: bool):
        """GEMM of the next block beside the epilogue/selection of the current one (default on); off = exclusive stage times."""
        L.check(L.lib().ldw_set_overlap(self._ctx, int(bool(on))))

    def set_fused(self, on: bool):
        """GEMM + MI epilogue as one kernel for every block with a bucket guess (default on); off = the two-kernel path."""
        L.check(L.lib().ldw_set_fused(self._ctx, int(bool(on))))

    def set_mixed(self, on: bool):
        """High-limb block GEMM + gathered low-limb GEMM for the listed units in speculative blocks (default on)."""
        L.check(L.lib().ldw_set_mixed(self._ctx, int(bool(on))))

    def set_path(self, mode: int):
        """Block-wide pass of the speculative blocks: 0 auto (default), 1 limb GEMM paths only, 2 approximate-GEMM path
        (one dual-digit int8 pass + exact class-wise popcount sums of the listed units) or an error."""
        L.check(L.lib().ldw_set_path(self._ctx, int(mode)))

    def set_select(self, mode: int):
        """Long-range selection: 0 auto (sort-free where it applies), 1 always the two radix sorts.  Same tables."""
        L.check(L.lib().ldw_set_select(self._ctx, int(mode)))

    def apx_info(self):
        """Diagnostics of the approximate path for the current weights (after set_weights)."""
        v = np.zeros(6)
        L.check(L.lib().ldw_apx_info(self._ctx, L.ptr(v)))
        return dict(usable=bool(v[0]), delta=float(v[1]), classes=int(v[2]), segments=int(v[3]), transitions=int(v[4]), e_last=int(v[5]))

    def set_screen(self, mode: int):
        """fp32 screen before the fp64 MI evaluation in speculative blocks: 0 off, 1 on (default), 2 verify."""
        L.check(L.lib().ldw_set_screen(self._ctx, int(mode)))

    def set_engine(self, engine: int):
        L.check(L.lib().ldw_set_engine(self._ctx, int(engine)))

    # -- alignment -----------------------------------------------------------
    def reserve(self, L_snps: int, N_seqs: int, max_blk_sz: int = 10000):
        """Start the side thread that creates the pass's streams, pinned staging buffers and code objects (ldw_ctx_reserve) — behind the
        upload of the alignment and the Hamming GEMM instead of inside the first pass.  Optional; set_alignment calls it."""
        L.check(L.lib().ldw_ctx_reserve(self._ctx, int(L_snps), int(N_seqs), int(max_blk_sz)))

    def set_alignment(self, states, max_blk_sz: int = 10000):
        """states: (L, N) uint8 numpy array or CUDA torch tensor (values 0..4)."""
        if _is_torch(states):
            assert states.dtype.__str__() == "torch.uint8" and states.dim() == 2 and states.is_contiguous()
            Ls, Ns = states.shape
            if states.is_cuda:
                # the library reads the tensor on ITS stream: make sure whatever torch stream produced it is done
                import torch
                torch.cuda.synchronize(states.device)
            L.check(L.lib().ldw_set_alignment(self._ctx, L.ptr(states), Ls, Ns, 1 if states.is_cuda else 0))
        else:
            st = L.as_c(states, np.uint8)
            assert st.ndim == 2
            Ls, Ns = st.shape
            L.check(L.lib().ldw_set_alignment(self._ctx, L.ptr(st), Ls, Ns, 0))
        self.L, self.N = int(Ls), int(Ns)
        if not getattr(self, "_reserved", False):
            self._reserved = True
            self.reserve(self.L, self.N, max_blk_sz)   # (a side thread: the pinned staging buffers, while the caller goes on to the Hamming weights)

    def alignment_scan(self, chars: np.ndarray) -> np.ndarray:
        """Upload the raw (N, L_total) alignment and return the 5 x L_total allele counts of every column."""
        ch = L.as_c(chars.view(np.uint8) if chars.dtype != np.uint8 else chars, np.uint8)
        out = np.empty((ch.shape[1], 5), dtype=np.int32)
        L.check(L.lib().ldw_alignment_scan(self._ctx, L.ptr(ch), ch.shape[0], ch.shape[1], L.ptr(out)))
        self._scanned = ch.shape
        return np.ascontiguousarray(out.T)

    def encode_alignment(self, chars, pos: np.ndarray, want_table=True, shape=None):
        """chars: (N, L_total) bytes, or None to reuse the alignment kept by ``alignment_scan``; pos: 1-based retained
        columns.  The result becomes the engine's alignment; returns ACGTN_table (5, n_pos)."""
        ps = L.as_c(pos, np.int32)
        tab = np.zeros((len(ps), 5), dtype=np.int32) if want_table else None
        if chars is None:
            n, lt = shape or self._scanned
            L.check(L.lib().ldw_encode_alignment(self._ctx, None, n, lt, L.ptr(ps), len(ps), L.ptr(tab)))
        else:
            ch = L.as_c(chars.view(np.uint8) if chars.dtype != np.uint8 else chars, np.uint8)
            n, lt = ch.shape
            L.check(L.lib().ldw_encode_alignment(self._ctx, L.ptr(ch), n, lt, L.ptr(ps), len(ps), L.ptr(tab)))
        self.L, self.N = len(ps), n
        return None if tab is None else np.ascontiguousarray(tab.T)

    def get_alignment(self) -> np.ndarray:
        out = np.empty((self.L, self.N), dtype=np.uint8)
        L.check(L.lib().ldw_get_alignment(self._ctx, L.ptr(out)))
        return out

    def state_counts(self) -> np.ndarray:
        out = np.empty((self.L, 5), dtype=np.int32)
        L.check(L.lib().ldw_state_counts(self._ctx, L.ptr(out)))
        return np.ascontiguousarray(out.T)  # 5 x L like ACGTN_table

    # -- Hamming weights -------------------------------------------------------
    def hamming_weights(self, thresh: int, want_shared=False):
        hdw = np.empty(self.N, dtype=np.float64)
        shared = np.empty((self.N, self.N), dtype=np.int32) if want_shared else None
        L.check(L.lib().ldw_hamming_weights(self._ctx, int(thresh), L.ptr(hdw), L.ptr(shared)))
        return (hdw, shared) if want_shared else hdw

    def hamming_counts(self, thresh: int, tile0: int, tile1: int) -> np.ndarray:
        """Contribution of the strip of 128-sequence row tiles [tile0, tile1) to the neighbour counts n_j (all j)."""
        out = np.zeros(self.N, dtype=np.int64)
        L.check(L.lib().ldw_hamming_counts(self._ctx, int(thresh), int(tile0), int(tile1), L.ptr(out)))
        return out

    # -- MI --------------------------------------------------------------------
    def set_weights(self, hdw, nlimbs: int = 0):
        w = L.as_c(hdw, np.float64)
        L.check(L.lib().ldw_set_weights(self._ctx, L.ptr(w), len(w), int(nlimbs)))

    def set_snp_meta(self, r, uqe, POS, paint, g):
        r_ = L.as_c(r, np.float64)
        uq = L.as_c(np.asarray(uqe) != 0, np.uint8)
        ps = L.as_c(POS, np.int32)
        pt = None if paint is None else L.as_c(paint, np.int32)
        assert r_.shape == (self.L,) and uq.shape == (self.L, 5) and ps.shape == (self.L,)
        L.check(L.lib().ldw_set_snp_meta(self._ctx, L.ptr(r_), L.ptr(uq), L.ptr(ps), L.ptr(pt), float(g)))

    def mi_block(self, from_idx, to_idx, quirk=L.QUIRK_REFERENCE, out=None) -> np.ndarray:
        """MI of one block as the reference's nf x nt matrix (Fortran order)."""
        fi = L.as_c(from_idx, np.int32)
        ti = L.as_c(to_idx, np.int32)
        if out is not None and _is_torch(out):
            L.check(L.lib().ldw_mi_block(self._ctx, L.ptr(fi), len(fi), L.ptr(ti), len(ti), quirk, L.ptr(out), 1))
            return out
        buf = np.empty(len(fi) * len(ti), dtype=np.float64)
        L.check(L.lib().ldw_mi_block(self._ctx, L.ptr(fi), len(fi), L.ptr(ti), len(ti), quirk, L.ptr(buf), 0))
        return buf.reshape((len(fi), len(ti)), order="F")

    def joint_tables(self, pair_a, pair_b):
        """(counts, fixed, frac_bits): int64 (P,5,5) unweighted joint counts and fixed-point weighted sums."""
        pa = L.as_c(pair_a, np.int32)
        pb = L.as_c(pair_b, np.int32)
        cnt = np.empty((len(pa), 5, 5), dtype=np.int64)
        fix = np.empty((len(pa), 5, 5), dtype=np.int64)
        fb = C.c_int(0)
        L.check(L.lib().ldw_joint_tables(self._ctx, L.ptr(pa), L.ptr(pb), len(pa), L.ptr(cnt), L.ptr(fix), C.byref(fb)))
        return cnt, fix, fb.value

    def mi_all_pairs(self, blocks, sr_dist=20000.0, lr_retain_links=1e6, lr_links_approx=1.0, sr_only=False,
                     quirk=L.QUIRK_REFERENCE, keep_sr=True):
        """blocks: (nb, 4) int32 rows (from_s, from_e, to_s, to_e), 1-based inclusive."""
        bl = L.as_c(blocks, np.int32).reshape(-1, 4)
        p = L.MIParams(float(sr_dist), float(lr_retain_links), float(lr_links_approx), int(bool(sr_only)), int(quirk),
                       int(bool(keep_sr)), 0)
        L.check(L.lib().ldw_mi_all_pairs(self._ctx, L.ptr(bl), len(bl), C.byref(p), 1))
        self._nblocks = len(bl)

    @staticmethod
    def mi_all_pairs_multi(engines, blocks, sr_dist=20000.0, lr_retain_links=1e6, lr_links_approx=1.0, sr_only=False,
                           quirk=L.QUIRK_REFERENCE, keep_sr=True, sr_rows_stay=False):
        """The block loop over several engines of THIS process, one per GPU (ldw_mi_all_pairs_multi): every engine must hold the same
        alignment, weights and SNP meta data; the blocks are dealt by cost, each engine runs its share on a worker thread inside the
        library, and engines[0] ends up with the assembled tables in make_blocks order (and the block statistics of all blocks) — exactly
        as if it had run every block itself.  Returns dict(owner=int32[nblocks], pass_ms, gather_ms, per_engine_ms).
        ``sr_rows_stay`` (LDW_MI_SR_ROWS_STAY): only the long-range table is assembled; every engine keeps the short-range rows of its own
        share and ``EngineGroup(engines)`` runs the short-range model over them."""
        bl = L.as_c(blocks, np.int32).reshape(-1, 4)
        p = L.MIParams(float(sr_dist), float(lr_retain_links), float(lr_links_approx), int(bool(sr_only)), int(quirk),
                       int(bool(keep_sr)), L.MI_SR_ROWS_STAY if sr_rows_stay else 0)
        arr = (C.c_void_p * len(engines))(*[e._ctx.value for e in engines])
        owner = np.zeros(len(bl), dtype=np.int32)
        ms = np.zeros(10, dtype=np.float64)
        L.check(L.lib().ldw_mi_all_pairs_multi(arr, len(engines), L.ptr(bl), len(bl), C.byref(p), L.ptr(owner), L.ptr(ms)))
        engines[0]._nblocks = len(bl)
        for k, e in enumerate(engines[1:], start=1):
            e._nblocks = int((owner == k).sum())
        return dict(owner=owner, pass_ms=float(ms[0]), gather_ms=float(ms[1]), per_engine_ms=ms[2:2 + min(len(engines), 8)].tolist())

    @staticmethod
    def hamming_weights_multi(engines, thresh: int) -> np.ndarray:
        """estimate_Hamming_distance_weights with the sequence x sequence comparison cut into one strip per engine (same alignment on all)."""
        arr = (C.c_void_p * len(engines))(*[e._ctx.value for e in engines])
        hdw = np.empty(engines[0].N, dtype=np.float64)
        L.check(L.lib().ldw_hamming_weights_multi(arr, len(engines), int(thresh), L.ptr(hdw)))
        return hdw

    def links_begin(self, nblocks: int):
        L.check(L.lib().ldw_links_begin(self._ctx, int(nblocks)))
        self._nblocks = 0

    def mi_block_links(self, from_idx, to_idx, sr_dist=20000.0, lr_retain_links=1e6, lr_links_approx=1.0, sr_only=False,
                       quirk=L.QUIRK_REFERENCE, keep_sr=True):
        fi = L.as_c(from_idx, np.int32)
        ti = L.as_c(to_idx, np.int32)
        p = L.MIParams(float(sr_dist), float(lr_retain_links), float(lr_links_approx), int(bool(sr_only)), int(quirk),
                       int(bool(keep_sr)), 0)
        L.check(L.lib().ldw_mi_block_links(self._ctx, L.ptr(fi), len(fi), L.ptr(ti), len(ti), C.byref(p)))
        self._nblocks += 1

    def links_end(self):
        L.check(L.lib().ldw_links_end(self._ctx))

    def links_count(self, which: int) -> int:
        n = C.c_int64(0)
        L.check(L.lib().ldw_links_count(self._ctx, int(which), C.byref(n)))
        return int(n.value)

    def links_view(self, which: int):
        """(a, b, MI) as torch tensors that ALIAS the context's own table in HBM (no copy): valid until the next call that
        changes the table; do not write to them.  STREAM ORDER: the library writes its tables from its own streams, so any
        asynchronous torch read of these views (a cat, a send) must have COMPLETED — synchronise the torch stream that reads —
        before the next ldw_mi_all_pairs / ldw_links_begin / ldw_links_import on this engine (dist.gather_begin does)."""
        import torch
        pa, pb, pm, n = C.c_void_p(0), C.c_void_p(0), C.c_void_p(0), C.c_int64(0)
        L.check(L.lib().ldw_links_device_ptrs(self._ctx, int(which), C.byref(pa), C.byref(pb), C.byref(pm), C.byref(n)))

        class _View:
            def __init__(self, ptr, n, typestr):
                self.__cuda_array_interface__ = dict(shape=(n,), typestr=typestr, data=(ptr or 0, False), version=2)

        dev = torch.device("cuda", self.device)
        if n.value == 0:
            return torch.empty(0, dtype=torch.int32, device=dev), torch.empty(0, dtype=torch.int32, device=dev), torch.empty(0, dtype=torch.float64, device=dev)
        mk = lambda p, ts: torch.as_tensor(_View(p.value, n.value, ts), device=dev)
        return mk(pa, "<i4"), mk(pb, "<i4"), mk(pm, "<f8")

    def links(self, which: int, device_tensors=False):
        """(a, b, MI): 0-based from-side / to-side SNP indices and MI of the short-range (0) or long-range (1) table."""
        n = self.links_count(which)
        if device_tensors:
            import torch
            dev = torch.device("cuda", self.device)
            a = torch.empty(n, dtype=torch.int32, device=dev)
            b = torch.empty(n, dtype=torch.int32, device=dev)
            mi = torch.empty(n, dtype=torch.float64, device=dev)
            L.check(L.lib().ldw_links_fetch(self._ctx, which, L.ptr(a), L.ptr(b), L.ptr(mi), n, 1))
            return a, b, mi
        a = np.empty(n, dtype=np.int32)
        b = np.empty(n, dtype=np.int32)
        mi = np.empty(n, dtype=np.float64)
        L.check(L.lib().ldw_links_fetch(self._ctx, which, L.ptr(a), L.ptr(b), L.ptr(mi), n, 0))
        return a, b, mi

    def links_import(self, which: int, a, b, mi):
        """Replace the context's sr (0) / lr (1) table, e.g. by the table assembled from all ranks (dist.gather_link_tables)."""
        if _is_torch(a):
            a, b, mi = a.contiguous(), b.contiguous(), mi.contiguous()
            assert a.dtype.__str__() == "torch.int32" and mi.dtype.__str__() == "torch.float64" and len(a) == len(b) == len(mi)
            on_dev = int(a.is_cuda)
            if on_dev:   # the library copies on its own (non-blocking) stream: torch's producers (cat, RCCL receives) must be done
                import torch
                torch.cuda.synchronize(a.device)
        else:
            a, b, mi = np.ascontiguousarray(a, dtype=np.int32), np.ascontiguousarray(b, dtype=np.int32), np.ascontiguousarray(mi, dtype=np.float64)
            on_dev = 0
        L.check(L.lib().ldw_links_import(self._ctx, int(which), L.ptr(a), L.ptr(b), L.ptr(mi), len(mi), on_dev))

    def block_stats(self):
        nb = self._nblocks
        t = np.empty(nb, dtype=np.int64)
        k = np.empty(nb, dtype=np.int64)
        s = np.empty(nb, dtype=np.int64)
        d = np.empty(nb, dtype=np.float64)
        L.check(L.lib().ldw_block_stats(self._ctx, nb, L.ptr(t), L.ptr(k), L.ptr(s), L.ptr(d)))
        return dict(n_lr_total=t, n_lr_kept=k, n_sr=s, disc_thresh=d)

    # -- short-range model / ARACNE on the device-resident sr table ---------------
    def sr_len_quantiles(self, nclust: int, sr_dist: float, prob: float = 0.95):
        """(q_lo, q_hi, n), each (nclust, S) with S = ceil(sr_dist)-1; column l-1 is len l."""
        S = int(np.ceil(sr_dist)) - 1
        qlo = np.empty((nclust, S), dtype=np.float64)
        qhi = np.empty((nclust, S), dtype=np.float64)
        n = np.empty((nclust, S), dtype=np.int64)
        L.check(L.lib().ldw_sr_len_quantiles(self._ctx, int(nclust), float(sr_dist), float(prob), S, L.ptr(qlo), L.ptr(qhi), L.ptr(n)))
        return qlo, qhi, n

    def sr_excess_stats(self, mean_dist: np.ndarray) -> np.ndarray:
        md = np.ascontiguousarray(mean_dist, dtype=np.float64)
        out = np.empty((md.shape[0], 5), dtype=np.float64)
        L.check(L.lib().ldw_sr_excess_stats(self._ctx, md.shape[0], md.shape[1], L.ptr(md), L.ptr(out)))
        return out

    def sr_pvalues(self, mean_dist: np.ndarray, shape: np.ndarray, srp_cutoff: float):
        """Returns (n_red, n_pool, min MI kept)."""
        md = np.ascontiguousarray(mean_dist, dtype=np.float64)
        sh = np.ascontiguousarray(shape, dtype=np.float64)
        assert sh.shape == (md.shape[0], 3)
        nr, npool, mn = C.c_int64(0), C.c_int64(0), C.c_double(0)
        L.check(L.lib().ldw_sr_pvalues(self._ctx, md.shape[0], md.shape[1], L.ptr(md), L.ptr(sh), float(srp_cutoff),
                                       C.byref(nr), C.byref(npool), C.byref(mn)))
        self._n_red, self._n_pool = nr.value, npool.value
        return nr.value, npool.value, mn.value

    def sr_reduced(self):
        n = self._n_red
        row = np.empty(n, dtype=np.int64)
        cc = np.empty(n, dtype=np.int32)
        first = np.empty(n, dtype=np.int32)
        dup = np.empty(n, dtype=np.uint8)
        srp = np.empty(n, dtype=np.float64)
        a = np.empty(n, dtype=np.int32)
        b = np.empty(n, dtype=np.int32)
        mi = np.empty(n, dtype=np.float64)
        L.check(L.lib().ldw_sr_reduced_fetch(self._ctx, n, L.ptr(row), L.ptr(a), L.ptr(b), L.ptr(mi), L.ptr(cc), L.ptr(first),
                                             L.ptr(dup), L.ptr(srp)))
        return dict(row=row, a=a, b=b, MI=mi, clust_c=cc, first_clust=first, dup=dup.astype(bool), srp_max=srp)

    def sr_pool(self):
        n = self._n_pool
        a = np.empty(n, dtype=np.int32)
        b = np.empty(n, dtype=np.int32)
        mi = np.empty(n, dtype=np.float64)
        L.check(L.lib().ldw_sr_pool_fetch(self._ctx, n, L.ptr(a), L.ptr(b), L.ptr(mi)))
        return a, b, mi

    # -- r05: the same model with the table left on the ranks that computed it (dist_srp.py) ---------
    def sr_excess_stats_blocks(self, mean_dist: np.ndarray, rows_per_block) -> np.ndarray:
        """(nblocks, nclust, 5): ldw_sr_excess_stats per reference block of this engine's table (rows of the blocks in order)."""
        md = np.ascontiguousarray(mean_dist, dtype=np.float64)
        rows = np.ascontiguousarray(rows_per_block, dtype=np.int64)
        out = np.zeros((len(rows), md.shape[0], 5), dtype=np.float64)
        L.check(L.lib().ldw_sr_excess_stats_blocks(self._ctx, md.shape[0], md.shape[1], L.ptr(md), len(rows), L.ptr(rows), L.ptr(out)))
        return out

    def sr_tail_extract(self, lower: np.ndarray, on_device: bool = False):
        """Rows at or above ``lower`` (nclust, S) per (cluster, len): (cnt, mi) with cnt (S, nclust) — len-major — and mi their MI values
        grouped in that order: a host array, or (``on_device``) a torch tensor on this engine's GPU — what an RCCL exchange sends as it is."""
        lo = np.ascontiguousarray(lower, dtype=np.float64)
        nclust, S = lo.shape
        cnt = np.zeros((S, nclust), dtype=np.int64)
        n = C.c_int64(0)
        L.check(L.lib().ldw_sr_tail_extract(self._ctx, nclust, S, L.ptr(lo), L.ptr(cnt), None, 0, 0, C.byref(n)))
        if on_device:
            import torch
            dev = torch.device("cuda", self.device)
            mi = torch.empty(n.value, dtype=torch.float64, device=dev)
            if n.value:
                torch.cuda.synchronize(dev)
                L.check(L.lib().ldw_sr_tail_extract(self._ctx, nclust, S, L.ptr(lo), L.ptr(cnt), L.ptr(mi), n.value, 1, C.byref(n)))
            return cnt, mi
        mi = np.empty(n.value, dtype=np.float64)
        if n.value:
            L.check(L.lib().ldw_sr_tail_extract(self._ctx, nclust, S, L.ptr(lo), L.ptr(cnt), L.ptr(mi), n.value, 0, C.byref(n)))
        return cnt, mi

    def sr_quantiles_merge(self, prob: float, cnts: list, mis: list, n_total: np.ndarray):
        """(q_lo, q_hi, violations) of the groups over all ranks from the candidates ``sr_tail_extract`` gave on each (``cnts[r]`` (S, nclust),
        ``mis[r]``) and the groups' global sizes ``n_total`` (nclust, S)."""
        nt = np.ascontiguousarray(n_total, dtype=np.int64)
        nclust, S = nt.shape
        cs = [np.ascontiguousarray(c, dtype=np.int64) for c in cnts]
        assert len(cs) == len(mis) >= 1 and all(c.shape == (S, nclust) for c in cs)
        on_dev = all(_is_torch(m) and m.is_cuda for m in mis)    # (what sr_tail_extract(.., on_device=True) gave and an RCCL gather delivered)
        if on_dev:
            import torch
            ms = [m.contiguous() for m in mis]
            assert all(str(m.dtype) == "torch.float64" and m.device.index == self.device for m in ms)
            torch.cuda.synchronize(ms[0].device)                 # the library reads them on its own stream
            pm = (C.c_void_p * len(ms))(*[m.data_ptr() if m.numel() else None for m in ms])
        else:
            ms = [np.ascontiguousarray(m.cpu().numpy() if _is_torch(m) else m, dtype=np.float64) for m in mis]
            pm = (C.c_void_p * len(ms))(*[m.ctypes.data if m.size else None for m in ms])
        pc = (C.c_void_p * len(cs))(*[c.ctypes.data for c in cs])
        qlo = np.empty((nclust, S), dtype=np.float64)
        qhi = np.empty((nclust, S), dtype=np.float64)
        viol = C.c_int64(0)
        L.check(L.lib().ldw_sr_quantiles_merge(self._ctx, nclust, S, float(prob), len(ms), pm, pc, L.ptr(nt), int(on_dev), L.ptr(qlo), L.ptr(qhi), C.byref(viol)))
        return qlo, qhi, int(viol.value)

    def sr_pvalues_local(self, mean_dist: np.ndarray, shape: np.ndarray, srp_cutoff: float):
        """ldw_sr_pvalues without the pool: (n_red, min MI kept on this engine — NaN when nothing was kept)."""
        md = np.ascontiguousarray(mean_dist, dtype=np.float64)
        sh = np.ascontiguousarray(shape, dtype=np.float64)
        assert sh.shape == (md.shape[0], 3)
        nr, mn = C.c_int64(0), C.c_double(0)
        L.check(L.lib().ldw_sr_pvalues(self._ctx, md.shape[0], md.shape[1], L.ptr(md), L.ptr(sh), float(srp_cutoff), C.byref(nr), None, C.byref(mn)))
        self._n_red, self._n_pool = nr.value, 0
        return nr.value, mn.value

    def sr_pool_build(self, min_mi: float) -> int:
        n = C.c_int64(0)
        L.check(L.lib().ldw_sr_pool_build(self._ctx, float(min_mi), C.byref(n)))
        self._n_pool = n.value
        return n.value

    def sr_reduced_import(self, a, b, mi, pool_a, pool_b, pool_mi):
        """Adopt the kept links and the pool of all ranks; ``aracne_device`` then answers for the kept links in the order given."""
        a, b, mi = L.as_c(a, np.int32), L.as_c(b, np.int32), L.as_c(mi, np.float64)
        pa, pb, pm = L.as_c(pool_a, np.int32), L.as_c(pool_b, np.int32), L.as_c(pool_mi, np.float64)
        assert len(a) == len(b) == len(mi) and len(pa) == len(pb) == len(pm)
        L.check(L.lib().ldw_sr_reduced_import(self._ctx, len(mi), L.ptr(a), L.ptr(b), L.ptr(mi), len(pm), L.ptr(pa), L.ptr(pb), L.ptr(pm)))
        self._n_red, self._n_pool = len(mi), len(pm)

    def aracne_device(self) -> np.ndarray:
        """ARACNE flags of the kept links (order of sr_reduced()) against the device-resident pool."""
        out = np.ones(self._n_red, dtype=np.uint8)
        L.check(L.lib().ldw_aracne_device(self._ctx, self._n_red, L.ptr(out)))
        return out.astype(bool)

    # -- consumers of the link tables ---------------------------
    def lr_tukey(self, min_links: int = 5000, sr=None):
        """Tukey thresholds of the long-range table, outlier links and ARACNE pool left on the device
        (analyse_long_range_links, R/lr_analyser.R:72-111).  ``sr`` = (a, b, MI) of the REDUCED short-range links — the rows of
        sr_links.tsv, which is what the reference pools with the long-range links (R/lr_analyser.R:67,106-109); None: no rows."""
        q13, thr = np.zeros(2), np.zeros(2)
        fb = C.c_int(0)
        nr, npool = C.c_int64(0), C.c_int64(0)
        if sr is None:
            sa = sb = smi = None
            ns = 0
        else:
            sa, sb, smi = L.as_c(sr[0], np.int32), L.as_c(sr[1], np.int32), L.as_c(sr[2], np.float64)
            ns = len(smi)
            assert len(sa) == len(sb) == ns
        L.check(L.lib().ldw_lr_tukey(self._ctx, int(min_links), L.ptr(sa), L.ptr(sb), L.ptr(smi), ns, L.ptr(q13), L.ptr(thr),
                                     C.byref(fb), C.byref(nr), C.byref(npool)))
        self._n_red, self._n_pool = nr.value, npool.value
        return dict(q13=q13, thresholds=thr, fallback=bool(fb.value), n_red=nr.value, n_pool=npool.value)

    def lr_reduced(self):
        n = self._n_red
        row = np.empty(n, dtype=np.int64)
        a = np.empty(n, dtype=np.int32)
        b = np.empty(n, dtype=np.int32)
        mi = np.empty(n, dtype=np.float64)
        L.check(L.lib().ldw_lr_reduced_fetch(self._ctx, n, L.ptr(row), L.ptr(a), L.ptr(b), L.ptr(mi)))
        return dict(row=row, a=a, b=b, MI=mi)

    def ldmap(self, reducer: int = 0, from_: int = 0, to: int = 0):
        """Block-summed, log-scaled, 0..1-rescaled LD map of all links (genomewide_LDMap, R/LDSummaryPlot.R:55-106).
        Returns (htm [B, B], n_pos, reducer)."""
        n_pos, r, B = C.c_int64(0), C.c_int32(0), C.c_int32(0)
        L.check(L.lib().ldw_ldmap(self._ctx, int(reducer), int(from_), int(to), C.byref(n_pos), C.byref(r), C.byref(B), None, 0))
        htm = np.empty((B.value, B.value), dtype=np.float64)
        L.check(L.lib().ldw_ldmap(self._ctx, int(reducer), int(from_), int(to), C.byref(n_pos), C.byref(r), C.byref(B), L.ptr(htm), htm.size))
        return htm, n_pos.value, r.value

    # -- element-wise twins ------------------------------------------------------
    def acgtn2num(self, nv: np.ndarray, ref_chars) -> None:
        """In-place twin of .ACGTN2num: nv is a Fortran-ordered (5, L) float64 matrix."""
        assert nv.dtype == np.float64 and nv.shape[0] == 5 and nv.flags.f_contiguous
        if isinstance(ref_chars, (bytes, bytearray)):
            ref = bytes(ref_chars)
        else:  # like as<char>(cv[c]): first character of each string ("" -> NUL, leaves the column untouched)
            ref = b"".join((bytes(s[:1]) if isinstance(s, (bytes, bytearray)) else str(s)[:1].encode("latin1")) or b"\0"
                           for s in ref_chars)
        assert len(ref) == nv.shape[1]
        buf = C.create_string_buffer(ref, len(ref))
        L.check(L.lib().ldw_acgtn2num(self._ctx, L.ptr(nv), C.cast(buf, C.c_void_p), nv.shape[1], 1))

    def fast_hadamard(self, MI, den, uq, pxy, pxpy, RXY, pXrX, pYrY) -> None:
        ops = [np.asarray(a, dtype=np.float64).reshape(-1, order="F") for a in (den, uq, pxy, pxpy, RXY, pXrX, pYrY)]
        flat = np.ascontiguousarray(MI.reshape(-1, order="F"))
        L.check(L.lib().ldw_fast_hadamard(self._ctx, L.ptr(flat), *[L.ptr(o) for o in ops], flat.size, 0))
        np.copyto(MI, flat.reshape(MI.shape, order="F"))


def aracne(chk_pos1, chk_pos2, chk_MI, full_pos1, full_pos2, full_MI) -> np.ndarray:
    c = [L.as_c(x, np.float64) for x in (chk_pos1, chk_pos2, chk_MI)]
    f = [L.as_c(x, np.float64) for x in (full_pos1, full_pos2, full_MI)]
    out = np.ones(len(c[0]), dtype=np.uint8)
    L.check(L.lib().ldw_aracne(None, L.ptr(c[0]), L.ptr(c[1]), L.ptr(c[2]), len(c[0]), L.ptr(f[0]), L.ptr(f[1]), L.ptr(f[2]),
                               len(f[0]), L.ptr(out)))
    return out.astype(bool)


def format_number(x: float) -> str:
    """One double as write.table prints it (native twin of rcompat.format_number)."""
    buf = C.create_string_buffer(64)
    L.check(L.lib().ldw_format_number(float(x), buf, 64))
    return buf.value.decode()


def write_table_tsv(path: str, columns, append: bool = True, nthreads: int = 0) -> int:
    """write.table(append = T, quote = F, row.names = F, col.names = F, sep = '\\t') of numeric columns by the native
    writer (R/computePairwiseMI.R:140,362): integer arrays print as integers, floating ones by R's 15-digit rule."""
    cols, kinds = [], []
    for c in columns:
        c = np.asarray(c)
        if c.dtype.kind in "iub":
            c = np.ascontiguousarray(c, dtype=np.int64)
            kinds.append(L.COL_INT64)
        else:
            c = np.ascontiguousarray(c, dtype=np.float64)
            kinds.append(L.COL_DOUBLE)
        cols.append(c)
    n = len(cols[0]) if cols else 0
    assert all(len(c) == n for c in cols)
    kind = np.asarray(kinds, dtype=np.int32)
    ptrs = (C.c_void_p * len(cols))(*[c.ctypes.data for c in cols])
    nb = C.c_int64(0)
    L.check(L.lib().ldw_write_table_tsv(str(path).encode(), int(bool(append)), n, len(cols), L.ptr(kind), C.cast(ptrs, C.c_void_p), int(nthreads),
                                        C.byref(nb)))
    return int(nb.value)



class EngineGroup:
    """The short-range model's view of SEVERAL engines of this process after ``Engine.mi_all_pairs_multi(.., sr_rows_stay=True)``: the reductions
    of ``srp.merge_n_sort_sr_links_device`` run over the engines' own rows inside the library (ldw_sr_len_quantiles_multi / _excess_stats_multi /
    _pvalues_multi: a worker thread per context; docs/HISTORY.md 7b), everything after them on engines[0]."""

    def __init__(self, engines):
        self.engines = list(engines)
        self._arr = (C.c_void_p * len(self.engines))(*[e._ctx.value for e in self.engines])
        self.device = self.engines[0].device

    def links_count(self, which: int) -> int:
        return self.engines[0].links_count(which) if which else sum(e.links_count(0) for e in self.engines)

    def sr_len_quantiles(self, nclust: int, sr_dist: float, prob: float = 0.95):
        S = int(np.ceil(sr_dist)) - 1
        qlo = np.empty((nclust, S), dtype=np.float64)
        qhi = np.empty((nclust, S), dtype=np.float64)
        n = np.empty((nclust, S), dtype=np.int64)
        L.check(L.lib().ldw_sr_len_quantiles_multi(self._arr, len(self.engines), int(nclust), float(sr_dist), float(prob), S, L.ptr(qlo), L.ptr(qhi), L.ptr(n)))
        return qlo, qhi, n

    def sr_excess_stats(self, mean_dist: np.ndarray) -> np.ndarray:
        md = np.ascontiguousarray(mean_dist, dtype=np.float64)
        out = np.empty((md.shape[0], 5), dtype=np.float64)
        L.check(L.lib().ldw_sr_excess_stats_multi(self._arr, len(self.engines), md.shape[0], md.shape[1], L.ptr(md), L.ptr(out)))
        return out

    def sr_pvalues(self, mean_dist: np.ndarray, shape: np.ndarray, srp_cutoff: float):
        md = np.ascontiguousarray(mean_dist, dtype=np.float64)
        sh = np.ascontiguousarray(shape, dtype=np.float64)
        assert sh.shape == (md.shape[0], 3)
        nr, npool, mn = C.c_int64(0), C.c_int64(0), C.c_double(0)
        L.check(L.lib().ldw_sr_pvalues_multi(self._arr, len(self.engines), md.shape[0], md.shape[1], L.ptr(md), L.ptr(sh), float(srp_cutoff),
                                             C.byref(nr), C.byref(npool), C.byref(mn)))
        self.engines[0]._n_red, self.engines[0]._n_pool = nr.value, npool.value
        return nr.value, npool.value, mn.value

    def sr_reduced(self):
        return self.engines[0].sr_reduced()

    def sr_pool(self):
        return self.engines[0].sr_pool()

    def aracne_device(self):
        return self.engines[0].aracne_device()
