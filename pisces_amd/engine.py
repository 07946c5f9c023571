"""Host-side mirror of the reference's operator surface for the hot path, over the C ABI.

`HipVariantCaller` plays the three objects Factory.CreateSomaticVariantCaller builds
(src/exe/Pisces/Logic/Factory.cs:253-269): ICandidateVariantFinder + IStateManager
(AddAlleleCounts / GetAlleleCount / GetCandidatesToProcess / DoneProcessing) + IAlleleCaller.Call,
with the reference's method names so the parity tests read like the reference's own tests.
Every compute call goes through libpisceship.so (HIP); nothing here has a CPU path.
"""
import ctypes as C

import numpy as np

from . import _abi
from ._native import PiscesHipError, lib


def _check(handle, rc):
    if rc != 0:
        msg = lib.pisces_hip_last_error(handle)
        raise PiscesHipError(rc, msg.decode() if msg else "")


class DeviceReadBatch:
    """A PiscesReadBatch whose arrays are torch tensors in device memory (what pisces_hip_add_device_reads takes): position / cigar_offset /
    seq_offset int32, flags / cigar_op / bases / quals (/ directions / deletion_directions) uint8, cigar_len int32 holding the uint32 bits."""

    def __init__(self, position, flags, cigar_offset, cigar_op, cigar_len, seq_offset, bases, quals, directions=None, deletion_directions=None,
                 n_ops=None, n_bases=None):
        import torch
        self.torch = torch
        self.tensors = dict(position=position, flags=flags, cigar_offset=cigar_offset, cigar_op=cigar_op, cigar_len=cigar_len, seq_offset=seq_offset,
                            bases=bases, quals=quals, directions=directions, deletion_directions=deletion_directions)
        for k, t in self.tensors.items():
            assert t is None or (t.is_cuda and t.is_contiguous()), k
        self.n_reads = int(position.numel())
        self.n_ops = int(cigar_op.numel()) if n_ops is None else int(n_ops)
        self.n_bases = int(bases.numel()) if n_bases is None else int(n_bases)
        c = _abi.PiscesReadBatch()
        c.n_reads = self.n_reads
        for k, t in self.tensors.items():
            setattr(c, k, C.cast(C.c_void_p(t.data_ptr() if t is not None else 0), type(getattr(c, k))))
        self.c = c

    @classmethod
    def from_host(cls, batch, device="cuda:0"):
        import torch
        up = lambda a, dt=None: None if a is None else torch.from_numpy(np.ascontiguousarray(a if dt is None else a.view(dt))).to(device)
        # (an empty tensor has no storage to point at: one spare element keeps every pointer valid)
        pad = lambda t: t if t is None or t.numel() else torch.zeros(1, dtype=t.dtype, device=device)
        return cls(pad(up(batch.position)), pad(up(batch.flags)), pad(up(batch.cigar_offset)), pad(up(batch.cigar_op)), pad(up(batch.cigar_len, np.int32)),
                   pad(up(batch.seq_offset)), pad(up(batch.bases)), pad(up(batch.quals)), pad(up(batch.directions)),
                   pad(up(getattr(batch, "deletion_directions", None))), n_ops=len(batch.cigar_op), n_bases=len(batch.bases))

    def synchronize(self):
        self.torch.cuda.synchronize(self.tensors["position"].device)


class _RowsAt:
    """memory of the library (the pinned download buffer) shown to numpy without a copy"""
    __slots__ = ("__array_interface__",)


class HipVariantCaller:
    def __init__(self, config=None, device=0):
        self.config = config if config is not None else _abi.default_config()
        h = C.c_void_p()
        rc = lib.pisces_hip_create(C.byref(self.config), device, C.byref(h))
        if rc != 0:
            raise PiscesHipError(rc, (lib.pisces_hip_last_error(None) or b"").decode())
        self._h = h
        self.device = device

    # ---- lifetime (C# IDisposable) ----
    def close(self):
        if getattr(self, "_h", None):
            self._torch_stream = None   # (an ExternalStream over the handle's stream: dead with the handle)
            lib.pisces_hip_destroy(self._h)
            self._h = None

    Dispose = close

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def handle(self):
        return self._h

    # ---- ChrReference ----
    def SetReference(self, sequence):
        """ChrReference.Sequence (upper case). position p is sequence[p-1]."""
        if isinstance(sequence, str):
            sequence = sequence.encode()
        arr = np.frombuffer(bytes(sequence), dtype=np.uint8) if not isinstance(sequence, np.ndarray) else \
            np.ascontiguousarray(sequence, np.uint8)
        _check(self._h, lib.pisces_hip_set_reference(self._h, arr.ctypes.data, arr.size))

    def SetIntervals(self, intervals):
        """ChrIntervalSet (sorted, disjoint, inclusive)."""
        iv = np.asarray(intervals, dtype=np.int32).reshape(-1, 2)   # (pairs, or an (n, 2) array: 16 000 tuples taken apart in Python were 2 ms a call)
        s, e = np.ascontiguousarray(iv[:, 0]), np.ascontiguousarray(iv[:, 1])
        _check(self._h, lib.pisces_hip_set_intervals(self._h, s.ctypes.data, e.ctypes.data, len(s)))

    def SetOwnedRange(self, lo, hi):
        """pisces_hip_set_owned_range: the positions this shard owns (candidates of its halo reads outside them are the neighbour's)."""
        _check(self._h, lib.pisces_hip_set_owned_range(self._h, int(lo), int(hi)))

    # ---- IStateManager ----
    def AddAlleleCounts(self, reads):
        """IStateManager.AddAlleleCounts for a batch (one _abi.ReadBatch, or an iterable of read dicts)."""
        batch = reads if isinstance(reads, _abi.ReadBatch) else _abi.ReadBatch(reads)
        _check(self._h, lib.pisces_hip_add_reads(self._h, C.byref(batch.c)))

    def AddDeviceReads(self, reads):
        """pisces_hip_add_device_reads: IStateManager.AddAlleleCounts for a batch that lies in device memory.  `reads`: a DeviceReadBatch
        (torch tensors on the handle's device), or an _abi.ReadBatch whose arrays are copied there first (test plumbing)."""
        b = reads if isinstance(reads, DeviceReadBatch) else DeviceReadBatch.from_host(reads if isinstance(reads, _abi.ReadBatch) else _abi.ReadBatch(reads),
                                                                                       f"cuda:{self.device}")
        b.synchronize()
        _check(self._h, lib.pisces_hip_add_device_reads(self._h, C.byref(b.c), b.n_ops, b.n_bases))

    def StageReads(self, reads):
        """pisces_hip_stage_reads: the arrays of `reads` (an _abi.ReadBatch) written into the handle's pinned staging buffer, as a host
        that marshals its reads straight into it would leave them; returns the batch to hand to AddAlleleCounts (valid until the next
        call on the handle that is not that AddAlleleCounts)."""
        b = reads if isinstance(reads, _abi.ReadBatch) else _abi.ReadBatch(reads)
        views = _abi.PiscesReadBatch()
        dd = getattr(b, "deletion_directions", None)
        _check(self._h, lib.pisces_hip_stage_reads(self._h, b.n_reads, len(b.cigar_op), b.n_bases, 1 if b.directions is not None else 0,
                                                   1 if dd is not None else 0, C.byref(views)))

        def fill(ptr, arr):
            if arr is not None and arr.size:
                C.memmove(ptr, arr.ctypes.data, arr.nbytes)

        fill(views.position, b.position); fill(views.flags, b.flags); fill(views.cigar_offset, b.cigar_offset)
        fill(views.cigar_op, b.cigar_op); fill(views.cigar_len, b.cigar_len); fill(views.seq_offset, b.seq_offset)
        fill(views.bases, b.bases); fill(views.quals, b.quals); fill(views.directions, b.directions); fill(views.deletion_directions, dd)
        staged = _abi.ReadBatch.__new__(_abi.ReadBatch)
        staged.c = views
        staged.n_reads = b.n_reads
        staged.n_bases = b.n_bases
        return staged

    def AddObservations(self, positions, tuples):
        positions = np.ascontiguousarray(positions, np.int32)
        tuples = np.ascontiguousarray(tuples, np.uint32)
        assert positions.shape == tuples.shape
        _check(self._h, lib.pisces_hip_add_observations(self._h, positions.ctypes.data, tuples.ctypes.data, positions.size))

    def GetCounts(self, start_position, n):
        out = np.zeros((n, 6, 3, _abi.NUM_ANCHORS), dtype=np.int32)
        _check(self._h, lib.pisces_hip_get_counts(self._h, start_position, n, out.ctypes.data))
        return out

    def GetBaseQualitySums(self, start_position, n):
        """RegionState._sumOfAlleleBaseQualities of [start_position, start_position + n): double[n][6][3][11]."""
        out = np.zeros((n, 6, 3, _abi.NUM_ANCHORS), dtype=np.float64)
        _check(self._h, lib.pisces_hip_get_base_quality_sums(self._h, int(start_position), int(n), out.ctypes.data))
        return out

    def GetGappedMnvRefCount(self, position):
        v = C.c_int32(0)
        _check(self._h, lib.pisces_hip_get_gapped_mnv_ref(self._h, int(position), C.byref(v)))
        return int(v.value)

    def GetAlleleCount(self, position, allele_type, direction_type, minAnchor=0, maxAnchor=None, fromEnd=False,
                       symmetric=False):
        """IAlleleSource.GetAlleleCount (src/lib/Pisces.Domain/Interfaces/IAlleleSource.cs): the anchor window
        arithmetic is AlleleCountHelper.GetAnchorAdjustedAlleleCount (AlleleCountHelper.cs:21-85) applied to
        the device-served counts."""
        if position <= 0:
            raise PiscesHipError(_abi.E_INVALID_ARG, "Position must be greater than 0.")
        c = self.GetCounts(position, 1)[0, allele_type, direction_type]
        return anchor_adjusted_count(c, minAnchor, maxAnchor, fromEnd, symmetric)

    def AddGappedMnvRefCount(self, support_lookup):
        pos = np.array(list(support_lookup.keys()), dtype=np.int32)
        cnt = np.array(list(support_lookup.values()), dtype=np.int32)
        _check(self._h, lib.pisces_hip_add_gapped_mnv_ref(self._h, pos.ctypes.data, cnt.ctypes.data, len(pos)))

    # ---- GetCandidatesToProcess + IAlleleCaller.Call + DoneProcessing ----
    def Call(self, upToPosition=None, capacity=1 << 16, reuse_buffer=False):
        """SmallVariantCaller.Call(upToPosition) (SmallVariantCaller.cs:157-189). None = final flush.
        Returns a CALLED_ALLELE_DTYPE array sorted by (position, ref, alt).  reuse_buffer: the rows are a view into a buffer the engine
        keeps from call to call (valid until the next Call), as a host that owns its output buffer works; the default hands
        out a fresh array every time."""
        up_to = -1 if upToPosition is None else int(upToPosition)
        while True:
            if reuse_buffer:
                out = getattr(self, "_flush_out", None)
                if out is None or len(out) < capacity:
                    out = self._flush_out = np.zeros(capacity, dtype=_abi.CALLED_ALLELE_DTYPE)
            else:
                out = np.zeros(capacity, dtype=_abi.CALLED_ALLELE_DTYPE)
            n = C.c_int64(0)
            rc = lib.pisces_hip_flush(self._h, up_to, out.ctypes.data, len(out), C.byref(n))
            if rc == _abi.E_BUFFER_TOO_SMALL:
                capacity = int(n.value)
                continue
            _check(self._h, rc)
            return out[: n.value]

    def CallView(self, upToPosition=None):
        """pisces_hip_flush_view: Call() whose rows are NOT copied — the array aliases memory of the handle (the pinned buffer the device
        wrote the records to) and is valid until the next Call* on this caller."""
        rows, n = C.c_void_p(), C.c_int64(0)
        _check(self._h, lib.pisces_hip_flush_view(self._h, -1 if upToPosition is None else int(upToPosition), C.byref(rows), C.byref(n), None, None, None, None, None))
        return self._rows_at(rows, n.value)

    def CallEndView(self):
        """pisces_hip_flush_end_view: CallEnd() without the copy (valid until the next Call* / CallBegin)."""
        rows, n = C.c_void_p(), C.c_int64(0)
        _check(self._h, lib.pisces_hip_flush_end_view(self._h, C.byref(rows), C.byref(n), None, None, None, None, None))
        return self._rows_at(rows, n.value)

    @staticmethod
    def _rows_at(rows, n):
        if not n:
            return np.zeros(0, dtype=_abi.CALLED_ALLELE_DTYPE)
        # (through the array interface: a ctypes array type per row count is ~5 us of Python a flush, this is ~2)
        w = _RowsAt()
        w.__array_interface__ = {"shape": (n,), "typestr": "|V%d" % _abi.CALLED_ALLELE_DTYPE.itemsize, "data": (rows.value, False), "version": 3}
        return np.asarray(w).view(_abi.CALLED_ALLELE_DTYPE)

    def CallBegin(self, upToPosition=None):
        """pisces_hip_flush_begin: the flush enqueued, DoneProcessing committed; the alleles come with CallEnd.  In between the next
        reads may be staged and added."""
        _check(self._h, lib.pisces_hip_flush_begin(self._h, -1 if upToPosition is None else int(upToPosition)))

    def CallEnd(self, capacity=1 << 16, reuse_buffer=False):
        """pisces_hip_flush_end: the alleles of the flush CallBegin started (the rows Call would have returned)."""
        while True:
            if reuse_buffer:
                out = getattr(self, "_flush_out", None)
                if out is None or len(out) < capacity:
                    out = self._flush_out = np.zeros(capacity, dtype=_abi.CALLED_ALLELE_DTYPE)
            else:
                out = np.zeros(capacity, dtype=_abi.CALLED_ALLELE_DTYPE)
            n = C.c_int64(0)
            rc = lib.pisces_hip_flush_end(self._h, out.ctypes.data, len(out), C.byref(n))
            if rc == _abi.E_BUFFER_TOO_SMALL:
                capacity = int(n.value)
                continue
            _check(self._h, rc)
            return out[: n.value]

    def CallWithAlleles(self, upToPosition=None, capacity=1 << 16):
        """Call() that also returns the (ref, alt) allele strings of every row: Reference / SNV rows from the record's
        base codes, insertion / deletion rows from the candidate the library found (pisces_hip_flush_ex)."""
        up_to = -1 if upToPosition is None else int(upToPosition)
        return self._rows_with_alleles(lambda *outputs: lib.pisces_hip_flush_ex(self._h, up_to, *outputs), capacity)

    def CallEndWithAlleles(self, capacity=1 << 16):
        """CallEnd() with the allele strings (pisces_hip_flush_end_ex): what CallWithAlleles would have returned for the flush
        CallBegin started."""
        return self._rows_with_alleles(lambda *outputs: lib.pisces_hip_flush_end_ex(self._h, *outputs), capacity)

    def _rows_with_alleles(self, entry, capacity):
        cand_cap, pool_cap = 1024, 1 << 16
        while True:
            out = np.zeros(capacity, dtype=_abi.CALLED_ALLELE_DTYPE)
            idx = np.zeros(capacity, dtype=np.int32)
            cands = (_abi.PiscesCandidate * cand_cap)()
            pool = np.zeros(pool_cap, dtype=np.uint8)
            n, nc, nb = C.c_int64(0), C.c_int64(0), C.c_int64(0)
            rc = entry(out.ctypes.data, capacity, C.byref(n), idx.ctypes.data, cands, cand_cap, C.byref(nc), pool.ctypes.data, pool_cap, C.byref(nb))
            if rc == _abi.E_BUFFER_TOO_SMALL:
                capacity = max(capacity, int(n.value))
                cand_cap = max(cand_cap, int(nc.value))
                pool_cap = max(pool_cap, int(nb.value))
                continue
            _check(self._h, rc)
            recs = out[: n.value]
            # (point rows in one pass over the array; only the rows that stand on a candidate are looked up one by one)
            letters = np.array(list(_abi.BASE_OF_ALLELE) + ["?", "?"])
            info = recs["info"].astype(np.int64)
            alleles = list(zip(letters[_abi.info_ref(info)].tolist(), letters[_abi.info_alt(info)].tolist()))
            for i in np.nonzero(idx[: n.value] >= 0)[0].tolist():
                c = cands[idx[i]]
                o = c.allele_offset
                alleles[i] = (bytes(pool[o: o + c.ref_len]).decode("latin-1"), bytes(pool[o + c.ref_len: o + c.ref_len + c.alt_len]).decode("latin-1"))
            return recs, alleles

    def GetCandidates(self, upToPosition=None):
        """The insertion / deletion candidates held by the state manager: list of dicts."""
        up_to = -1 if upToPosition is None else int(upToPosition)
        n, nb = C.c_int64(0), C.c_int64(0)
        _check(self._h, lib.pisces_hip_get_candidates(self._h, up_to, None, 0, C.byref(n), None, 0, C.byref(nb)))
        cands = (_abi.PiscesCandidate * max(1, n.value))()
        pool = np.zeros(max(1, nb.value), dtype=np.uint8)
        _check(self._h, lib.pisces_hip_get_candidates(self._h, up_to, cands, n.value, C.byref(n), pool.ctypes.data, nb.value, C.byref(nb)))
        out = []
        for i in range(n.value):
            c = cands[i]
            o = c.allele_offset
            out.append({"position": c.position, "category": c.category, "ref": bytes(pool[o: o + c.ref_len]).decode("latin-1"),
                        "alt": bytes(pool[o + c.ref_len: o + c.ref_len + c.alt_len]).decode("latin-1"),
                        "support_by_dir": list(c.support_by_dir), "well_anchored_by_dir": list(c.well_anchored_by_dir),
                        "open_left": bool(c.open_left), "open_right": bool(c.open_right)})
        return out

    @staticmethod
    def _candidate_arrays(items):
        """[(position, ref, alt)] or [dict(position, category, ref, alt, support_by_dir, ...)] -> (PiscesCandidate[], allele pool)."""
        arr = (_abi.PiscesCandidate * max(1, len(items)))()
        pool = bytearray()
        for i, it in enumerate(items):
            d = it if isinstance(it, dict) else {"position": it[0], "ref": it[1], "alt": it[2]}
            c = arr[i]
            c.position, c.category = int(d["position"]), int(d.get("category", 0))
            c.ref_len, c.alt_len = len(d["ref"]), len(d["alt"])
            for k in range(3):
                c.support_by_dir[k] = int(d.get("support_by_dir", (0, 0, 0))[k])
                c.well_anchored_by_dir[k] = int(d.get("well_anchored_by_dir", (0, 0, 0))[k])
            c.open_left, c.open_right = int(bool(d.get("open_left", False))), int(bool(d.get("open_right", False)))
            c.allele_offset = len(pool)
            pool += d["ref"].encode() + d["alt"].encode()
        return arr, np.frombuffer(bytes(pool) or b"\0", dtype=np.uint8).copy(), len(pool)

    def AddCandidates(self, candidates):
        """IStateManager.AddCandidates for candidates the caller brings itself (dicts as GetCandidates returns them)."""
        arr, pool, nb = self._candidate_arrays(list(candidates))
        _check(self._h, lib.pisces_hip_add_candidates(self._h, arr, len(candidates), pool.ctypes.data, nb))

    def SetForcedAlleles(self, alleles):
        """-forcedalleles: [(position, ref, alt)] of this chromosome (AlleleCaller.AddForcedGtAlleles + SmallVariantCaller's forced
        candidates).  After SetIntervals, before the first Call."""
        arr, pool, nb = self._candidate_arrays(list(alleles))
        _check(self._h, lib.pisces_hip_set_forced_alleles(self._h, arr, len(alleles), pool.ctypes.data, nb))

    def SetExactTotalNumCalled(self, on=True):
        """pisces_hip_set_exact_total_called: TotalNumCalled also counts the callable SNVs outside the interval set (AlleleCaller.cs:109-131)."""
        _check(self._h, lib.pisces_hip_set_exact_total_called(self._h, int(bool(on))))

    def SetKnownVariants(self, variants):
        """The chromosome's known (prior) variants, [(position, ref, alt)]: Factory.cs:204 hands them to VariantCollapser (AnnotateKnown,
        VariantCollapser.cs:178-190).  [] clears."""
        arr, pool, nb = self._candidate_arrays(list(variants))
        _check(self._h, lib.pisces_hip_set_known_variants(self._h, arr, len(variants), pool.ctypes.data, nb))

    def SetExcludeMNVsFromCollapsing(self, on=True):
        """PiscesApplicationOptions.ExcludeMNVsFromCollapsing (VariantCollapser.cs:33): MNV candidates are no targets of the collapser."""
        _check(self._h, lib.pisces_hip_set_exclude_mnvs_from_collapsing(self._h, int(bool(on))))

    # ---- multi-GPU summary (one process per GPU): RCCL bound at run time by the library ----
    @staticmethod
    def comm_unique_id():
        """128-byte RCCL id made by rank 0 and handed to the other processes."""
        buf = (C.c_uint8 * 128)()
        rc = lib.pisces_hip_comm_unique_id(buf, 128)
        if rc != 0:
            raise PiscesHipError(rc, (lib.pisces_hip_last_error(None) or b"").decode(errors="replace"))
        return bytes(buf)

    def comm_init(self, unique_id, rank, world):
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        _check(self._h, lib.pisces_hip_comm_init(self._h, buf, int(rank), int(world)))

    @staticmethod
    def comm_library():
        """pisces_hip_comm_library: '<source>: <path>' of the librccl the library binds (source = PISCES_HIP_RCCL_PATH | mapped | default)."""
        buf = C.create_string_buffer(4096)
        rc = lib.pisces_hip_comm_library(buf, len(buf))
        if rc < 0:
            raise PiscesHipError(rc, (lib.pisces_hip_last_error(None) or b"").decode(errors="replace"))
        return buf.value.decode()

    def comm_ranks(self):
        """ncclCommCount of the handle's communicator (1 without one)."""
        n = C.c_int32(0)
        _check(self._h, lib.pisces_hip_comm_ranks(self._h, C.byref(n)))
        return int(n.value)

    def reduce_summary(self, values):
        """In-place sum over the ranks of the int64[4] summary (identity without a communicator)."""
        v = (C.c_int64 * 4)(*[int(x) for x in values])
        _check(self._h, lib.pisces_hip_reduce_summary(self._h, v))
        return [int(x) for x in v]

    def FindCandidates(self, reads, snvs_and_mnvs=True, call_mnvs=False, max_mnv_length=3, max_gap_between_mnv=1):
        """ICandidateVariantFinder.FindCandidates on the device (pisces_hip_find_candidates_device: the kernels pisces_hip_add_reads
        enqueues), against the reference given to SetReference: list of dicts per read event, unmerged, in read order."""
        batch = reads if isinstance(reads, _abi.ReadBatch) else _abi.ReadBatch(reads)
        cap, pool_cap = 4096, 1 << 18
        while True:
            cands = (_abi.PiscesCandidate * cap)()
            pool = np.zeros(pool_cap, dtype=np.uint8)
            nb = C.c_int64(0)
            n = lib.pisces_hip_find_candidates_device(self._h, C.byref(batch.c), int(snvs_and_mnvs), int(call_mnvs), max_mnv_length,
                                                      max_gap_between_mnv, cands, cap, pool.ctypes.data, pool_cap, C.byref(nb))
            if n == _abi.E_BUFFER_TOO_SMALL:
                cap *= 4
                pool_cap = max(pool_cap * 4, int(nb.value))
                continue
            if n < 0:
                _check(self._h, int(n))
            return _candidate_dicts(cands, pool, n)

    def Stats(self):
        s = (C.c_int64 * 4)()
        _check(self._h, lib.pisces_hip_stats(self._h, s))
        # SmallVariantCaller.cs:114-115 / AlignmentsSource.cs:63: the vector a multi-GPU job adds up over its interval shards
        return {"TotalNumCalled": s[0], "TotalNumCollapsed": s[1], "reads": s[2], "reads_skipped": s[3]}

    def HostTime(self, reset=False):
        """pisces_hip_host_time: where the host's time went inside the streaming surface."""
        t = (C.c_double * 4)()
        _check(self._h, lib.pisces_hip_host_time(self._h, t, 1 if reset else 0))
        flushes = int(t[3])
        return {"add_reads_s": t[0], "flush_s": t[1], "flush_wait_s": t[2], "flushes": flushes,
                "host_ms_per_flush": (t[1] - t[2]) / flushes * 1e3 if flushes else 0.0}

    def TransferBytes(self, reset=False):
        """pisces_hip_transfer_bytes: what crossed PCIe for this caller since the last reset."""
        b = (C.c_int64 * 4)()
        _check(self._h, lib.pisces_hip_transfer_bytes(self._h, b, 1 if reset else 0))
        return {"h2d_reads": int(b[0]), "d2h_records": int(b[1]), "d2h_candidates": int(b[2]), "d2h_counts": int(b[3])}

    # ---- device-resident surface ----
    def stream_handle(self):
        """pisces_hip_get_stream: the handle's own hipStream_t as an int (what stream=None stands for in the calls below)."""
        s = C.c_void_p()
        _check(self._h, lib.pisces_hip_get_stream(self._h, C.byref(s)))
        return int(s.value or 0)

    def torch_stream(self):
        """The handle's own stream as a torch.cuda.ExternalStream.  Tensors a torch host hands to the device-resident surface are
        allocated / filled under `with torch.cuda.stream(caller.torch_stream()):` so that torch's fill kernels and the library's launches
        sit in ONE queue: the handle's stream is hipStreamNonBlocking and does not order against torch's default (null) stream."""
        ts = getattr(self, "_torch_stream", None)
        if ts is None:
            import torch
            ts = self._torch_stream = torch.cuda.ExternalStream(self.stream_handle(), device=torch.device("cuda", self.device))
        return ts

    @staticmethod
    def _stream_arg(stream):
        """None = the handle's own stream (an explicit choice); a torch stream or a raw hipStream_t otherwise.  The null stream (0, e.g.
        torch.cuda.current_stream().cuda_stream of a default torch context) is refused: the C ABI reads NULL as "the handle's stream", which
        does not order against the null stream, so work torch queued there (a torch.zeros fill) would race the launch."""
        if stream is None:
            return None
        raw = int(getattr(stream, "cuda_stream", stream))
        if raw == 0:
            raise ValueError("stream 0 is HIP's null stream, which the handle's non-blocking stream is not ordered against: pass None (the handle's "
                             "own stream, after synchronizing your fills) or allocate under caller.torch_stream() and pass that")
        return raw

    def call_tiles(self, d_tuples, d_tiles, n_tiles, d_ref, ref_start, ref_len, d_records, capacity, d_tile_results, stream=None):
        """All pointer arguments are raw device addresses (ints); capacity >= 256 * n_tiles record slots."""
        _check(self._h, lib.pisces_hip_call_tiles(self._h, d_tuples, d_tiles, n_tiles, d_ref, ref_start, ref_len,
                                                  d_records, capacity, d_tile_results, self._stream_arg(stream)))

    def balanced_tile_loci(self, n_loci):
        """Tile size (<= 64) that gives every CU the same number of tiles for a launch over n_loci loci (pisces_hip_balanced_tile_loci)."""
        return int(lib.pisces_hip_balanced_tile_loci(self._h, int(n_loci)))

    def call_tiles_batched(self, batches, stream=None):
        """pisces_hip_call_tiles_batched: batches = list of (d_tuples, d_tiles, n_tiles, d_ref, ref_start, ref_len, d_records, capacity,
        d_tile_results) with per-batch output buffers; spread over the handle's lanes.  Waits for `stream` first when given; the outputs
        are complete after synchronize()."""
        arr = (_abi.PiscesTileBatch * max(len(batches), 1))()
        for i, (tu, ti, n, rf, rs, rl, rec, cap, tr) in enumerate(batches):
            arr[i].d_tuples, arr[i].d_tiles, arr[i].n_tiles, arr[i].ref_start_position = tu, ti, n, rs
            arr[i].d_ref_bases, arr[i].ref_length, arr[i].d_records, arr[i].d_tile_results, arr[i].record_capacity = rf, rl, rec, tr, cap
        _check(self._h, lib.pisces_hip_call_tiles_batched(self._h, arr, len(batches), self._stream_arg(stream)))

    def call_tiles_graph_build(self, batches):
        """pisces_hip_call_tiles_graph_build: the launches of `batches` (as call_tiles_batched takes them), in order, as one HIP graph."""
        arr = (_abi.PiscesTileBatch * max(len(batches), 1))()
        for i, (tu, ti, n, rf, rs, rl, rec, cap, tr) in enumerate(batches):
            arr[i].d_tuples, arr[i].d_tiles, arr[i].n_tiles, arr[i].ref_start_position = tu, ti, n, rs
            arr[i].d_ref_bases, arr[i].ref_length, arr[i].d_records, arr[i].d_tile_results, arr[i].record_capacity = rf, rl, rec, tr, cap
        gid = C.c_int32(-1)
        _check(self._h, lib.pisces_hip_call_tiles_graph_build(self._h, arr, len(batches), C.byref(gid)))
        return gid.value

    def call_tiles_graph_launch(self, graph_id, stream=None):
        _check(self._h, lib.pisces_hip_call_tiles_graph_launch(self._h, int(graph_id), self._stream_arg(stream)))

    def compact_records(self, d_records, d_tile_results, n_tiles, d_offsets, d_out, out_capacity, d_count, stream=None):
        _check(self._h, lib.pisces_hip_compact_records(self._h, d_records, d_tile_results, n_tiles, d_offsets, d_out,
                                                       out_capacity, d_count, self._stream_arg(stream)))

    def accumulate_tiles(self, d_tuples, d_tiles, n_tiles, d_counts, stream=None):
        _check(self._h, lib.pisces_hip_accumulate_tiles(self._h, d_tuples, d_tiles, n_tiles, d_counts, self._stream_arg(stream)))

    def device_totals(self, reset=False):
        """{records, candidate_loci, called (IAlleleCaller.TotalNumCalled), tiles} summed over call_tiles launches."""
        s = (C.c_int64 * 4)()
        _check(self._h, lib.pisces_hip_device_totals(self._h, s, 1 if reset else 0))
        return {"records": s[0], "candidate_loci": s[1], "called": s[2], "tiles": s[3]}

    def mark(self, which, stream=None):
        """pisces_hip_mark: an event on the launch stream in front of (0) / behind (1) a run of launches."""
        _check(self._h, lib.pisces_hip_mark(self._h, int(which), self._stream_arg(stream)))

    def marked_ms(self):
        ms = C.c_float(0)
        _check(self._h, lib.pisces_hip_marked_ms(self._h, C.byref(ms)))
        return ms.value

    def set_timing(self, enable=True):
        """enable = True / n > 0: time every launch / every n-th launch with HIP events; False / 0: off (default)."""
        _check(self._h, lib.pisces_hip_set_timing(self._h, int(enable)))

    def kernel_time(self):
        """(total_ms, launches) of the kernels launched since set_timing(True), from HIP events on the launch stream."""
        ms, n = C.c_double(0), C.c_int64(0)
        _check(self._h, lib.pisces_hip_kernel_time(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def SetChainTiming(self, enable=True):
        """pisces_hip_set_chain_timing: events around what AddDeviceReads enqueues and around a flush's kernels (up to the compacted records)."""
        _check(self._h, lib.pisces_hip_set_chain_timing(self._h, int(bool(enable))))

    def ChainTime(self):
        """(add_ms, flush_ms) of the last AddDeviceReads and the last flush since SetChainTiming(True): device time by HIP events."""
        out = (C.c_double * 2)()
        _check(self._h, lib.pisces_hip_chain_time(self._h, out))
        return out[0], out[1]

    def probe_read_bandwidth(self, nbytes=1 << 30, reps=6):
        """GB/s of a plain streaming read on this device (context for the roofline fraction)."""
        g = C.c_double(0)
        _check(self._h, lib.pisces_hip_probe_read_bandwidth(self._h, nbytes, reps, C.byref(g)))
        return g.value

    def synchronize(self):
        _check(self._h, lib.pisces_hip_synchronize(self._h))

    def last_kernel_ms(self):
        ms = C.c_float(0)
        _check(self._h, lib.pisces_hip_last_kernel_ms(self._h, C.byref(ms)))
        return ms.value

    def bgzf_inflate(self, file_bytes, check_crc=True):
        """Row f4 (upstream): every BGZF block of `file_bytes` (a BAM file or a region of one) inflated on the device, one wave per
        block (BamReader.ReadBlock -> UncompressBlock, BamReader.cs:603-645).  Returns (inflated bytes, block table, kernel ms)."""
        data = np.frombuffer(bytes(file_bytes), dtype=np.uint8)
        blocks, total = bgzf_scan(data)
        out = np.zeros(max(total, 1), dtype=np.uint8)
        ms = C.c_float(0)
        _check(self._h, lib.pisces_hip_bgzf_inflate(self._h, data.ctypes.data, data.size, blocks, len(blocks), out.ctypes.data, total,
                                                    1 if check_crc else 0, C.byref(ms)))
        return out[:total].tobytes(), blocks, ms.value

    def bam_decode(self, file_bytes, ref_id, min_map_quality=1, skip_duplicates=True, only_proper_pairs=False):
        """Row f4: a BAM file (bytes, or the array bam_stage returned) inflated and cut into records on the device; the alignments of
        reference sequence `ref_id` that AlignmentSource.ShouldSkipRead keeps become a device-resident read batch.  Returns {reads,
        skipped, cigar_ops, bases}."""
        data = file_bytes if isinstance(file_bytes, np.ndarray) else np.frombuffer(bytes(file_bytes), dtype=np.uint8)
        blocks, _ = bgzf_scan(data)
        counts = (C.c_int64 * 4)()
        _check(self._h, lib.pisces_hip_bam_decode(self._h, data.ctypes.data, data.size, blocks, len(blocks), int(ref_id), int(min_map_quality),
                                                  int(bool(skip_duplicates)), int(bool(only_proper_pairs)), counts))
        self._bam_counts = {"reads": counts[0], "skipped": counts[1], "cigar_ops": counts[2], "bases": counts[3]}
        mode = lib.pisces_hip_bam_chain_mode(self._h)
        return dict(self._bam_counts, chain="guessed" if mode == 0 else "hopped")

    def bam_fetch_directions(self):
        """(directions, deletion_directions) of the decoded batch — made when some read of it carries the Stitcher's XD tag — or None."""
        n = self._bam_counts
        dirs, dd = np.zeros(n["bases"], np.uint8), np.zeros(2 * n["cigar_ops"], np.uint8)
        rc = lib.pisces_hip_bam_fetch_directions(self._h, dirs.ctypes.data, dd.ctypes.data)
        if rc < 0:
            _check(self._h, rc)
        return (dirs, dd) if rc == 1 else None

    def bam_fetch(self):
        """The decoded batch as host arrays (dict with the PiscesReadBatch field names)."""
        n = self._bam_counts
        out = {"position": np.zeros(n["reads"], np.int32), "flags": np.zeros(n["reads"], np.uint8),
               "cigar_offset": np.zeros(n["reads"] + 1, np.int32), "cigar_op": np.zeros(n["cigar_ops"], np.uint8),
               "cigar_len": np.zeros(n["cigar_ops"], np.uint32), "seq_offset": np.zeros(n["reads"] + 1, np.int32),
               "bases": np.zeros(n["bases"], np.uint8), "quals": np.zeros(n["bases"], np.uint8)}
        _check(self._h, lib.pisces_hip_bam_fetch(self._h, *[out[k].ctypes.data for k in ("position", "flags", "cigar_offset", "cigar_op",
                                                                                       "cigar_len", "seq_offset", "bases", "quals")]))
        return out

    def AddDecodedReads(self):
        """AddAlleleCounts + FindCandidates for the batch bam_decode left on the device (bases and qualities never come back)."""
        _check(self._h, lib.pisces_hip_add_decoded_reads(self._h))


def device_count():
    """pisces_hip_device_count: the HIP devices of this process (a -threadbychr host gives job j the device j % count)."""
    n = lib.pisces_hip_device_count()
    if n < 0:
        raise PiscesHipError(n, "pisces_hip_device_count: no HIP runtime / device")
    return int(n)


def anchor_adjusted_count(c, minAnchor=0, maxAnchor=None, fromEnd=False, symmetric=False):
    """AlleleCountHelper.GetAnchorAdjustedAlleleCount (AlleleCountHelper.cs:21-85) over one [11] anchor row."""
    well = _abi.ANCHOR_SIZE
    n = _abi.NUM_ANCHORS
    true_min = min(well, minAnchor)
    init_max = well
    if maxAnchor is not None:
        init_max = well - 1 if maxAnchor >= well else maxAnchor
    tot = 0
    if fromEnd:
        for i in range(true_min, init_max + 1):
            tot += int(c[n - i - 1])
        if maxAnchor is None:
            for i in range(true_min if symmetric else 0, init_max):
                tot += int(c[i])
    else:
        for i in range(true_min, init_max + 1):
            tot += int(c[i])
        if maxAnchor is None:
            for i in range(init_max + 1, (n - true_min) if symmetric else n):
                tot += int(c[i])
    return tot


def expand_reads(batch, min_base_call_quality=20):
    """Host expansion of reads into (position, tuple) observations (pisces_hip_expand_reads)."""
    cap = batch.n_bases * 2 + 64
    while True:
        pos = np.zeros(cap, dtype=np.int32)
        tup = np.zeros(cap, dtype=np.uint32)
        n = lib.pisces_hip_expand_reads(C.byref(batch.c), min_base_call_quality, pos.ctypes.data, tup.ctypes.data, cap)
        if n == _abi.E_BUFFER_TOO_SMALL:
            cap *= 4
            continue
        if n < 0:
            raise PiscesHipError(int(n), "expand_reads failed")
        return pos[:n], tup[:n]


def new_pad_state():
    """Cursors of a VCF writer + RegionMapper pair at the start of a chromosome (PiscesVcfPadState)."""
    return _abi.PiscesVcfPadState(0, 0, -1)


def format_vcf(chrom, records, vcf_config=None, alleles=None, pad=None, **overrides):
    """VCF body lines of `records` (pisces_hip_format_vcf[_padded]).  alleles: the (ref, alt) string pairs CallWithAlleles returned, needed
    for insertion / deletion / MNV rows; vcf_config: _abi.PiscesVcfConfig or None for the defaults (+ field overrides, e.g. crush=1).
    pad: dict(state=new_pad_state(), reference=bytes, intervals=[(start, end), ...], finish=bool) adds RegionMapper's no-call rows for
    uncovered interval positions; the state object is advanced in place."""
    cfg = vcf_config
    if cfg is None:
        cfg = _abi.PiscesVcfConfig()
        _check(None, lib.pisces_hip_vcf_default_config(C.byref(cfg)))
    for k, v in overrides.items():
        setattr(cfg, k, v)
    recs = np.ascontiguousarray(records)
    n = len(recs)
    idx = cands = pool_arr = None
    if alleles is not None:
        idx_l, cand_l, pool = [], [], bytearray()
        for (r, a) in alleles:
            if len(r) == 1 and len(a) == 1:
                idx_l.append(-1)
                continue
            c = _abi.PiscesCandidate()
            c.ref_len, c.alt_len, c.allele_offset = len(r), len(a), len(pool)
            pool += r.encode() + a.encode()
            idx_l.append(len(cand_l))
            cand_l.append(c)
        idx = np.array(idx_l, dtype=np.int32)
        cands = (_abi.PiscesCandidate * max(len(cand_l), 1))(*cand_l)
        pool_arr = np.frombuffer(bytes(pool) + b"\0", dtype=np.uint8).copy()
    refa = starts = ends = state = None
    n_iv = finish = 0
    if pad is not None:
        refa = np.ascontiguousarray(np.frombuffer(pad["reference"], dtype=np.uint8) if isinstance(pad["reference"], (bytes, bytearray))
                                    else pad["reference"], np.uint8)
        starts = np.array([a for a, _ in pad["intervals"]], dtype=np.int32)
        ends = np.array([b for _, b in pad["intervals"]], dtype=np.int32)
        n_iv, finish, state = len(starts), int(bool(pad.get("finish"))), pad["state"]
    cap = 256 * max(n, 1) + 128 * (int((ends - starts + 1).sum()) if pad is not None and n_iv else 0)
    while True:
        buf = C.create_string_buffer(max(cap, 1))
        need = lib.pisces_hip_format_vcf_padded(
            C.byref(cfg), chrom.encode(), recs.ctypes.data if n else None, n, idx.ctypes.data if idx is not None else None, cands,
            pool_arr.ctypes.data if pool_arr is not None else None, refa.ctypes.data if refa is not None else None,
            refa.size if refa is not None else 0, starts.ctypes.data if n_iv else None, ends.ctypes.data if n_iv else None, n_iv,
            C.byref(state) if state is not None else None, finish, buf, cap)
        if need < 0:
            raise PiscesHipError(int(need), "format_vcf failed")
        if need <= cap:
            return buf.raw[:need].decode()
        cap = int(need)


def find_indel_candidates(batch, ref, min_base_call_quality=20):
    """Host finder for insertions / deletions (pisces_hip_find_indel_candidates): list of dicts in read order."""
    return find_candidates(batch, ref, min_base_call_quality, snvs_and_mnvs=False)


def bgzf_scan(file_bytes):
    """pisces_hip_bgzf_scan: the BGZF block table of the file bytes (host); returns (ctypes array of PiscesBgzfBlock, inflated size)."""
    data = file_bytes if isinstance(file_bytes, np.ndarray) else np.frombuffer(bytes(file_bytes), dtype=np.uint8)
    total = C.c_int64(0)
    n = lib.pisces_hip_bgzf_scan(data.ctypes.data, data.size, None, 0, C.byref(total))
    if n < 0:
        raise PiscesHipError(int(n), "not a chain of BGZF blocks")
    blocks = (_abi.PiscesBgzfBlock * max(int(n), 1))()
    n2 = lib.pisces_hip_bgzf_scan(data.ctypes.data, data.size, blocks, n, C.byref(total))
    assert n2 == n
    return (_abi.PiscesBgzfBlock * int(n)).from_buffer(blocks) if n else (_abi.PiscesBgzfBlock * 0)(), int(total.value)


def _candidate_dicts(cands, pool, n):
    out = []
    for i in range(n):
        c = cands[i]
        o = c.allele_offset
        out.append({"position": c.position, "category": c.category, "ref": bytes(pool[o: o + c.ref_len]).decode("latin-1"),   # (inserted bases are whatever bytes the read holds)
                    "alt": bytes(pool[o + c.ref_len: o + c.ref_len + c.alt_len]).decode("latin-1"),
                    "support_by_dir": list(c.support_by_dir), "well_anchored_by_dir": list(c.well_anchored_by_dir),
                    "open_left": bool(c.open_left), "open_right": bool(c.open_right)})
    return out


def find_candidates(batch, ref, min_base_call_quality=20, snvs_and_mnvs=True, call_mnvs=False, max_mnv_length=3, max_gap_between_mnv=1):
    """The host finder (pisces_hip_find_candidates): insertions / deletions, and with snvs_and_mnvs the SNV / MNV candidates of the
    M operations; list of dicts in read order."""
    refa = np.ascontiguousarray(np.frombuffer(ref, dtype=np.uint8) if isinstance(ref, (bytes, bytearray)) else ref, np.uint8)
    cap, pool_cap = 4096, 1 << 18
    while True:
        cands = (_abi.PiscesCandidate * cap)()
        pool = np.zeros(pool_cap, dtype=np.uint8)
        nb = C.c_int64(0)
        n = lib.pisces_hip_find_candidates(C.byref(batch.c), refa.ctypes.data, refa.size, min_base_call_quality, int(snvs_and_mnvs),
                                           int(call_mnvs), max_mnv_length, max_gap_between_mnv, cands, cap, pool.ctypes.data, pool_cap,
                                           C.byref(nb))
        if n == _abi.E_BUFFER_TOO_SMALL:
            cap *= 4
            pool_cap = max(pool_cap * 4, int(nb.value))
            continue
        if n < 0:
            raise PiscesHipError(int(n), "find_candidates failed")
        return _candidate_dicts(cands, pool, n)
