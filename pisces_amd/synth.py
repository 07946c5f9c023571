"""Deterministic synthetic amplicon pileups (SURVEY.md §8d recipe; generator modelled on the
reference's AmpliconTestFactory, src/test/TestUtilities/AmpliconTestFactory.cs:110-147).

One pileup = a seeded reference + `depth` reads per 150-locus amplicon, half forward / half
reverse by read-index parity, Q37 with a 2 % fraction at Q12 (exercises the N path), 0.1 %
uniform base errors, planted SNVs every 100th locus with VAF ~ U[0.02, 0.5] and strand ratio
~ U[0.3, 0.7].  Mismatches at the first/last read base and next to a low-quality base are
suppressed so every SNV candidate is fully anchored (the reference's collapser is then the
identity).

The same matrices give (a) device-resident tile-bucketed observation tuples for the HIP path and
(b) a `ReadBatch` of the first amplicons for the CPU oracle / streaming surface.
Runs on torch CPU or GPU tensors; all randomness comes from one seeded torch.Generator.
"""
import math
from dataclasses import dataclass

import numpy as np
import torch

from . import _abi

READ_LEN = 150
TILE = 64
_BASE_ASCII = np.frombuffer(b"AGCT", dtype=np.uint8)   # index = AlleleType code (A0 G1 C2 T3)


CHUNK_AMPLICONS = 32    # generation unit: every chunk of amplicons draws from its own seeded stream, so that any locus range of a
                        # (possibly very large) global pileup can be made without making the rest — what an interval shard does


@dataclass
class Pileup:
    n_loci: int
    depth: int
    region_start: int            # 1-based position of locus 0 of THIS pileup (= flank + 1 + first_locus)
    ref: torch.Tensor            # uint8 ASCII; ref[i] is position ref_start + i (the whole contig when the pileup starts at locus 0)
    tuples: torch.Tensor         # int32 view of packed uint32 tuples, tile-bucketed, each segment padded to x4
    tiles: torch.Tensor          # uint8 bytes of PiscesTile[n_tiles]
    n_tiles: int
    n_obs: int                   # real (unpadded) observations
    base: torch.Tensor           # (A, depth, READ_LEN) uint8 AlleleType code of every read base, amplicons first_amplicon ..
    qual: torch.Tensor           # (A, depth, READ_LEN) uint8
    planted: np.ndarray          # loci (relative to region_start) with a planted SNV
    ref_start: int = 1           # 1-based position of ref[0]
    first_locus: int = 0         # global locus index of locus 0 of this pileup
    first_amplicon: int = 0      # global index of base[0] / qual[0]
    total_loci: int = 0          # loci of the global pileup this one is a range of
    flank: int = READ_LEN

    @property
    def ref_len(self):
        return int(self.ref.numel())


def _amplicon_lengths(n_loci):
    a = math.ceil(n_loci / READ_LEN)
    lens = [READ_LEN] * a
    lens[-1] = n_loci - (a - 1) * READ_LEN
    return a, lens


def _chunk_generator(seed, chunk, dev):
    g = torch.Generator(device=dev)
    g.manual_seed((int(seed) * 1_000_003 + int(chunk) + 1) % (2 ** 63))
    return g


def make_pileup(n_loci, depth, seed=20260928, device="cpu", p_lowq=0.02, base_error=0.001, snv_every=100,
                snv_offset=37, q_hi=37, q_lo=12, vaf_range=(0.02, 0.5), strand_range=(0.3, 0.7), flank=READ_LEN, tile=TILE,
                first_locus=0, total_loci=None, with_tuples=True, tile_plan=None):
    """Loci [first_locus, first_locus + n_loci) of a global pileup of `total_loci` loci (default: the whole pileup, n_loci from 0).
    `tile` = loci per tile of the bucketed tuple stream (<= 64; the kernels take any PiscesTile.n_loci <= 64).  Every chunk of
    CHUNK_AMPLICONS amplicons has its own seeded stream: a range made on its own equals the same range of the whole.
    `tile_plan` = [(tiles, loci per tile), ...] for the first tiles of the stream (the rest take `tile`): tiles of unequal size."""
    assert 1 <= tile <= TILE
    dev = torch.device(device)
    total_loci = first_locus + n_loci if total_loci is None else int(total_loci)
    assert 0 <= first_locus and first_locus + n_loci <= total_loci and n_loci > 0
    A_total, lens_total = _amplicon_lengths(total_loci)
    a0, a1 = first_locus // READ_LEN, (first_locus + n_loci - 1) // READ_LEN          # amplicons this range touches
    ga0, ga1 = max(a0 - 1, 0), min(a1 + 1, A_total - 1)                                # + a neighbour each side for the reference margin
    c0, c1 = ga0 // CHUNK_AMPLICONS, ga1 // CHUNK_AMPLICONS
    lut = torch.from_numpy(_BASE_ASCII.copy()).to(dev)
    low_t, hi_t = torch.tensor(q_lo, device=dev, dtype=torch.uint8), torch.tensor(q_hi, device=dev, dtype=torch.uint8)
    reverse = (torch.arange(depth, device=dev) % 2 == 1).view(1, depth, 1)
    idx = torch.arange(READ_LEN, device=dev).view(1, 1, READ_LEN)
    refc_parts, base_parts, qual_parts = [], [], []
    for c in range(c0, c1 + 1):
        g = _chunk_generator(seed, c, dev)
        ca0 = c * CHUNK_AMPLICONS
        A = min(CHUNK_AMPLICONS, A_total - ca0)
        shape = (A, depth, READ_LEN)
        n_chunk = A * READ_LEN
        ref_codes = torch.randint(0, 4, (n_chunk,), generator=g, device=dev, dtype=torch.int64)
        refc = ref_codes.view(A, 1, READ_LEN)                                         # code of the reference base per locus
        lowq = torch.rand(shape, generator=g, device=dev) < p_lowq
        locus = (ca0 * READ_LEN + torch.arange(n_chunk, device=dev)).view(A, 1, READ_LEN)    # global locus index
        planted_mask = (locus % snv_every == snv_offset) & (locus < total_loci)
        # neighbours of planted sites are never low quality (keeps the planted candidates fully anchored)
        near_planted = torch.zeros_like(planted_mask)
        near_planted[..., 1:] |= planted_mask[..., :-1]
        near_planted[..., :-1] |= planted_mask[..., 1:]
        lowq &= ~near_planted
        qual = torch.where(lowq, low_t, hi_t)
        # per-site VAF / strand split / alt base
        site_vaf = vaf_range[0] + (vaf_range[1] - vaf_range[0]) * torch.rand((A, 1, READ_LEN), generator=g, device=dev)
        site_s = strand_range[0] + (strand_range[1] - strand_range[0]) * torch.rand((A, 1, READ_LEN), generator=g, device=dev)
        site_alt = (refc + 1 + torch.randint(0, 3, (A, 1, READ_LEN), generator=g, device=dev)) % 4
        p_alt = torch.where(reverse, site_vaf * 2 * (1 - site_s), site_vaf * 2 * site_s).clamp(max=1.0)
        u = torch.rand(shape, generator=g, device=dev)
        err_alt = (refc + 1 + torch.randint(0, 3, shape, generator=g, device=dev)) % 4
        is_err = u < base_error
        # suppress sequencing errors at the read ends and next to low-quality bases
        rlen = torch.tensor(lens_total[ca0:ca0 + A], device=dev).view(A, 1, 1)
        at_end = (idx == 0) | (idx == rlen - 1)
        nb_lowq = torch.zeros_like(lowq)
        nb_lowq[..., 1:] |= lowq[..., :-1]
        nb_lowq[..., :-1] |= lowq[..., 1:]
        is_err &= ~at_end & ~nb_lowq & ~planted_mask
        base = torch.where(is_err, err_alt, refc.expand(shape))
        base = torch.where(planted_mask & (u < p_alt), site_alt.expand(shape), base).to(torch.uint8)
        lo_a, hi_a = max(ga0, ca0) - ca0, min(ga1, ca0 + A - 1) - ca0 + 1            # the amplicons of this chunk that are wanted
        refc_parts.append(ref_codes[lo_a * READ_LEN: hi_a * READ_LEN])
        ka0, ka1 = max(a0, ca0) - ca0, min(a1, ca0 + A - 1) - ca0 + 1
        if ka1 > ka0:
            base_parts.append(base[ka0:ka1])
            qual_parts.append(qual[ka0:ka1])
        del lowq, u, err_alt, is_err, base, qual
    base = torch.cat(base_parts) if len(base_parts) > 1 else base_parts[0]
    qual = torch.cat(qual_parts) if len(qual_parts) > 1 else qual_parts[0]
    A = a1 - a0 + 1
    lens = lens_total[a0:a1 + 1]

    # reference: the flank before locus 0 and after the last locus from their own stream; positions: locus L <-> flank + 1 + L
    gf = _chunk_generator(seed, -1, dev)
    flank_codes = torch.randint(0, 4, (2 * flank,), generator=gf, device=dev, dtype=torch.int64)
    ref_codes = torch.cat(refc_parts)                                                  # loci [ga0 * READ_LEN, (ga1 + 1) * READ_LEN)
    ref_lo_locus = ga0 * READ_LEN
    ref_hi_locus = min((ga1 + 1) * READ_LEN, total_loci)
    ref_codes = ref_codes[: ref_hi_locus - ref_lo_locus]
    pieces, ref_start = [ref_codes], flank + 1 + ref_lo_locus
    if ga0 == 0:
        pieces.insert(0, flank_codes[:flank])
        ref_start = 1
    if ga1 == A_total - 1:
        pieces.append(flank_codes[flank:])
    ref_ascii = lut[torch.cat(pieces)].contiguous()
    region_start = flank + 1 + first_locus

    # anchor bin of read index i in a read of length rlen (GetAnchorType, RegionStateManager.cs:83-116)
    rlen = torch.tensor(lens, device=dev).view(A, 1, 1)
    left = idx.expand(A, 1, READ_LEN)
    right = (rlen - 1 - idx)
    anchor = torch.where(left >= right,
                         torch.where(right >= _abi.ANCHOR_SIZE, torch.tensor(_abi.ANCHOR_SIZE, device=dev), _abi.NUM_ANCHORS - right - 1),
                         torch.where(left >= _abi.ANCHOR_SIZE, torch.tensor(_abi.ANCHOR_SIZE, device=dev), left))
    direction = reverse.to(torch.int64)   # Forward 0 / Reverse 1
    shape = (A, depth, READ_LEN)
    # PISCES_TUPLE_PACK without the column (the locus-in-tile enters per tile below)
    packed = ((direction << 8) | (base.to(torch.int64) << 10) | (anchor.to(torch.int64) << 13) | (qual.to(torch.int64) << 24))
    packed = packed.expand(shape)
    dir_bit4 = (direction & 1) << 4

    if not with_tuples:   # reads only (reads_of / mixed_reads): no packed tuple stream
        gl = np.arange(first_locus, first_locus + n_loci)
        return Pileup(n_loci=n_loci, depth=depth, region_start=region_start, ref=ref_ascii, tuples=None, tiles=None, n_tiles=0, n_obs=0,
                      base=base, qual=qual, planted=np.nonzero((gl % snv_every) == snv_offset)[0], ref_start=ref_start,
                      first_locus=first_locus, first_amplicon=a0, total_loci=total_loci, flank=flank)
    # tile-bucketed tuple stream: tiles from locus 0 of this pileup; inside a tile read-major (each read's run of loci)
    edges = [0]
    for cnt_, loci_ in (tile_plan or ()):
        assert 1 <= loci_ <= TILE
        for _ in range(cnt_):
            if edges[-1] < n_loci:
                edges.append(min(edges[-1] + loci_, n_loci))
    while edges[-1] < n_loci:
        edges.append(min(edges[-1] + tile, n_loci))
    n_tiles = len(edges) - 1
    tiles = np.zeros(n_tiles, dtype=_abi.TILE_DTYPE)
    segs = []
    cursor = 0
    pad_val = torch.tensor([-1], device=dev, dtype=torch.int64)   # 0xFFFFFFFF after the int32 cast
    for t in range(n_tiles):
        l0, l1 = edges[t], edges[t + 1]                            # loci of this pileup
        g0, g1 = first_locus + l0, first_locus + l1               # global loci
        n_seg = 0
        for a in range(g0 // READ_LEN, (g1 - 1) // READ_LEN + 1):
            i0, i1 = max(g0, a * READ_LEN) - a * READ_LEN, min(g1, a * READ_LEN + lens_total[a]) - a * READ_LEN
            if i1 <= i0:
                continue
            loc = torch.arange(a * READ_LEN + i0 - g0, a * READ_LEN + i1 - g0, device=dev, dtype=torch.int64).view(1, -1)
            col = (((loc >> 2) & 15) | ((loc & 3) << 4)) ^ dir_bit4[0]          # PISCES_TUPLE_COLUMN(locus, direction)
            piece = (packed[a - a0, :, i0:i1] | (col << 2)).reshape(-1)
            segs.append(piece)
            n_seg += piece.numel()
        tiles[t] = (region_start + l0, l1 - l0, cursor, cursor + n_seg)
        padn = (-n_seg) % 4
        if padn:
            segs.append(pad_val.expand(padn))
        cursor += n_seg + padn
    tuples64 = torch.cat(segs) if segs else torch.zeros(0, dtype=torch.int64, device=dev)
    tuples = (tuples64 & 0xFFFFFFFF).to(torch.int64)
    tuples = torch.where(tuples >= 2 ** 31, tuples - 2 ** 32, tuples).to(torch.int32).contiguous()
    n_obs = int(sum(int(tiles[t]["tuple_end"] - tiles[t]["tuple_begin"]) for t in range(n_tiles)))
    tiles_t = torch.from_numpy(tiles.view(np.uint8).copy()).to(dev)
    gl = np.arange(first_locus, first_locus + n_loci)
    planted = np.nonzero((gl % snv_every) == snv_offset)[0]
    return Pileup(n_loci=n_loci, depth=depth, region_start=region_start, ref=ref_ascii, tuples=tuples, tiles=tiles_t,
                  n_tiles=n_tiles, n_obs=n_obs, base=base, qual=qual, planted=planted, ref_start=ref_start, first_locus=first_locus,
                  first_amplicon=a0, total_loci=total_loci, flank=flank)


def reference_of(total_loci, seed, flank=READ_LEN, device="cpu"):
    """The whole contig of a global pileup (position p = result[p - 1]) without making the pileup: the reference bases are the first
    draw of every chunk's stream.  `device` must be the device the pileups are made on (CPU and GPU generators differ)."""
    A_total, _ = _amplicon_lengths(total_loci)
    dev = torch.device(device)
    parts = []
    for c in range((A_total + CHUNK_AMPLICONS - 1) // CHUNK_AMPLICONS):
        A = min(CHUNK_AMPLICONS, A_total - c * CHUNK_AMPLICONS)
        parts.append(torch.randint(0, 4, (A * READ_LEN,), generator=_chunk_generator(seed, c, dev), device=dev, dtype=torch.int64))
    codes = torch.cat(parts)[:total_loci]
    fl = torch.randint(0, 4, (2 * flank,), generator=_chunk_generator(seed, -1, dev), device=dev, dtype=torch.int64)
    return _BASE_ASCII[torch.cat([fl[:flank], codes, fl[flank:]]).cpu().numpy()]


def reads_of(p, n_amplicons=None, first_amplicon=None):
    """ReadBatch of `n_amplicons` amplicons starting at the GLOBAL amplicon index `first_amplicon` (default: the pileup's first; all of
    them when n_amplicons is None): one <len>M read per row.  An amplicon the pileup's locus range only touches comes whole."""
    A = p.base.shape[0]
    first_amplicon = p.first_amplicon if first_amplicon is None else first_amplicon
    k0 = min(max(first_amplicon - p.first_amplicon, 0), A)
    n_amp = A - k0 if n_amplicons is None else min(A - k0, n_amplicons)
    _, lens_total = _amplicon_lengths(p.total_loci or (p.first_locus + p.n_loci))
    lens = lens_total[p.first_amplicon + k0:]
    base = p.base[k0:k0 + n_amp].cpu().numpy()
    qual = p.qual[k0:k0 + n_amp].cpu().numpy()
    depth = p.depth
    origin = p.flank + 1   # position of global locus 0
    pos, flags, cig_len, seq_off = [], [], [], [0]
    bases, quals = [], []
    for a in range(n_amp):
        L = lens[a]
        pos.append(np.full(depth, origin + (p.first_amplicon + k0 + a) * READ_LEN, dtype=np.int32))
        flags.append((np.arange(depth) % 2).astype(np.uint8))
        cig_len.append(np.full(depth, L, dtype=np.uint32))
        bases.append(_BASE_ASCII[base[a, :, :L]].reshape(-1))
        quals.append(qual[a, :, :L].reshape(-1))
        seq_off.extend(seq_off[-1] + L * (np.arange(depth) + 1))
    n = n_amp * depth
    return _abi.ReadBatch.from_arrays(
        position=np.concatenate(pos), flags=np.concatenate(flags), cigar_offset=np.arange(n + 1, dtype=np.int32),
        cigar_op=np.full(n, ord("M"), dtype=np.uint8), cigar_len=np.concatenate(cig_len),
        seq_offset=np.array(seq_off, dtype=np.int32), bases=np.concatenate(bases), quals=np.concatenate(quals))


def observations_of(p, n_tiles=None, first_tile=0):
    """(positions, tuples) numpy arrays of `n_tiles` tiles from `first_tile` on (default: the first `n_tiles`), padding removed, tile
    order kept."""
    tiles = p.tiles.cpu().numpy().view(_abi.TILE_DTYPE)
    first_tile = max(0, min(int(first_tile), p.n_tiles))
    nt = p.n_tiles if n_tiles is None else min(p.n_tiles, first_tile + n_tiles)
    if nt <= first_tile:
        return np.zeros(0, np.int32), np.zeros(0, np.uint32)
    lo, hi = int(tiles[first_tile]["tuple_begin"]), int(tiles[nt - 1]["tuple_end"])   # only the tiles asked for leave the device
    tup = p.tuples[lo:hi].cpu().numpy().view(np.uint32)
    pos_out, tup_out = [], []
    for t in range(first_tile, nt):
        b, e = int(tiles[t]["tuple_begin"]) - lo, int(tiles[t]["tuple_end"]) - lo
        seg = tup[b:e]
        pos_out.append((tiles[t]["start_position"] + _abi.tuple_fields(seg)[0]).astype(np.int32))
        tup_out.append(seg)
    return np.concatenate(pos_out), np.concatenate(tup_out)


# ---- BASELINE config 3: SNV + MNV + small indels ---------------------------------------------------------------
MNV_OFFSET, DEL_OFFSET, INS_OFFSET = 40, 60, 80   # read index of the planted event inside its amplicon


def mixed_event_of(a, sparse=1):
    """The event planted in (global) amplicon `a`: None, or (kind, offset, length, vaf) with kind 'M' (an MNV of 2-3 bases), 'D' (a
    deletion of 1-10 bases) or 'I' (an insertion of 1-6 bases).  One amplicon in 13 carries each kind: about one MNV, one deletion
    and one insertion per 1950 loci; every locus keeps the SNVs and errors of make_pileup.  sparse = 10: a tenth of that density
    (BASELINE config 4)."""
    if a % sparse:
        return None
    a //= sparse
    k = a % 13
    vaf = 0.05 + 0.30 * ((a * 2654435761) % 1000) / 1000.0
    if k == 3:
        return ("M", MNV_OFFSET, 2 + a % 2, vaf)
    if k == 7:
        return ("D", DEL_OFFSET, 1 + (a * 7) % 10, vaf)
    if k == 11:
        return ("I", INS_OFFSET, 1 + (a * 5) % 6, vaf)
    return None


def mixed_reads(p, seed=0, sparse=1, kinds="MDI"):
    """The reads of pileup `p` (whole amplicons only) with mixed_event_of's events planted: MNV carriers get the variant bases, deletion
    carriers the CIGAR xM dD yM, insertion carriers xM iI yM (inserted bases at Q37).  Returns (ReadBatch, planted) where planted is a
    list of (kind, position, ref, alt) in VCF form (anchor base included for insertions / deletions).  kinds: the event kinds to plant."""
    A = p.base.shape[0]
    _, lens_total = _amplicon_lengths(p.total_loci or (p.first_locus + p.n_loci))
    base = p.base.cpu().numpy()
    qual = p.qual.cpu().numpy()
    ref = p.ref.cpu().numpy()
    origin = p.flank + 1
    depth = p.depth
    rev = (np.arange(depth) % 2).astype(np.uint8)
    pos_l, flag_l, cop_l, clen_l, ncig_l, seq_l, q_l, slen_l, planted = [], [], [], [], [], [], [], [], []
    code_of = {ord("A"): 0, ord("G"): 1, ord("C"): 2, ord("T"): 3}
    for k in range(A):
        a = p.first_amplicon + k
        L = lens_total[a]
        start = origin + a * READ_LEN
        rows = _BASE_ASCII[base[k, :, :L]]            # (depth, L) ASCII
        qs = qual[k, :, :L]
        ev = mixed_event_of(a, sparse) if L == READ_LEN else None
        if ev is not None and ev[0] not in kinds:
            ev = None
        rng = np.random.default_rng((seed * 1_000_003 + a) % (2 ** 63))
        carriers = np.zeros(depth, dtype=bool)
        if ev is not None:
            carriers = rng.random(depth) < ev[3]
        refrow = ref[start - p.ref_start: start - p.ref_start + L]
        if ev is not None and ev[0] == "M":
            off, n = ev[1], ev[2]
            alt = _BASE_ASCII[(np.array([code_of[int(c)] for c in refrow[off:off + n]]) + 1 + np.arange(n)) % 4]
            rows = rows.copy()
            qs = qs.copy()
            rows[np.ix_(carriers, np.arange(off, off + n))] = alt
            qs[np.ix_(carriers, np.arange(off - 1, off + n + 1))] = 37     # clean flanks: the MNV is fully anchored in every carrier
            planted.append(("M", start + off, bytes(refrow[off:off + n]).decode(), bytes(alt).decode()))
        if ev is None or ev[0] == "M":
            pos_l.append(np.full(depth, start, np.int32)); flag_l.append(rev)
            cop_l.append(np.full(depth, ord("M"), np.uint8)); clen_l.append(np.full(depth, L, np.uint32)); ncig_l.append(np.ones(depth, np.int64))
            seq_l.append(rows.reshape(-1)); q_l.append(qs.reshape(-1)); slen_l.append(np.full(depth, L, np.int64))
            continue
        off, n = ev[1], ev[2]
        nc = int(carriers.sum())
        keep = ~carriers
        # non-carriers: <L>M
        pos_l.append(np.full(depth, start, np.int32)); flag_l.append(np.concatenate([rev[keep], rev[carriers]]))
        ops = np.full(int(keep.sum()) + 3 * nc, ord("M"), np.uint8)
        lens = np.full(int(keep.sum()) + 3 * nc, L, np.uint32)
        if ev[0] == "D":
            ops[int(keep.sum()) + 1::3] = ord("D")
            lens[int(keep.sum())::3] = off; lens[int(keep.sum()) + 1::3] = n; lens[int(keep.sum()) + 2::3] = L - off - n
            crow = np.concatenate([rows[carriers][:, :off], rows[carriers][:, off + n:]], axis=1)
            cq = np.concatenate([qs[carriers][:, :off], qs[carriers][:, off + n:]], axis=1)
            cq[:, off - 1:off + 1] = 37                                     # both flanks pass CheckDeletionQuality
            planted.append(("D", start + off - 1, bytes(refrow[off - 1:off + n]).decode(), chr(refrow[off - 1])))
            clen = L - n
        else:
            ops[int(keep.sum()) + 1::3] = ord("I")
            lens[int(keep.sum())::3] = off; lens[int(keep.sum()) + 1::3] = n; lens[int(keep.sum()) + 2::3] = L - off
            ins = _BASE_ASCII[rng.integers(0, 4, n)]
            crow = np.concatenate([rows[carriers][:, :off], np.tile(ins, (nc, 1)), rows[carriers][:, off:]], axis=1)
            cq = np.concatenate([qs[carriers][:, :off], np.full((nc, n), 37, np.uint8), qs[carriers][:, off:]], axis=1)
            planted.append(("I", start + off - 1, chr(refrow[off - 1]), chr(refrow[off - 1]) + bytes(ins).decode()))
            clen = L + n
        cop_l.append(ops); clen_l.append(lens)
        ncig_l.append(np.concatenate([np.ones(int(keep.sum()), np.int64), np.full(nc, 3, np.int64)]))
        seq_l.append(np.concatenate([rows[keep].reshape(-1), crow.reshape(-1)]))
        q_l.append(np.concatenate([qs[keep].reshape(-1), cq.reshape(-1)]))
        slen_l.append(np.concatenate([np.full(int(keep.sum()), L, np.int64), np.full(nc, clen, np.int64)]))
    ncig = np.concatenate(ncig_l)
    slen = np.concatenate(slen_l)
    batch = _abi.ReadBatch.from_arrays(
        position=np.concatenate(pos_l), flags=np.concatenate(flag_l), cigar_offset=np.concatenate([[0], np.cumsum(ncig)]).astype(np.int32),
        cigar_op=np.concatenate(cop_l), cigar_len=np.concatenate(clen_l), seq_offset=np.concatenate([[0], np.cumsum(slen)]).astype(np.int32),
        bases=np.concatenate(seq_l), quals=np.concatenate(q_l))
    return batch, planted
