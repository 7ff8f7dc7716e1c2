"""Doc-range sharded build: the exchange step between GPUs.

Each rank builds the index of its document range with GLOBAL coordinates (`Builder(coord_base=...)`); the exchange --
per-term segments grouped by owner, heavy terms balanced by posting units, all-to-all-v of the three export buffers over RCCL /
xGMI, merge at the owner (IndexSequence.Shift + AddRange semantics of MergeFiles, Index.cs:665-674) -- happens INSIDE the library
(round 4: dcd_comm_create, dcd_build_merge_distributed, dcd_index_allgather, dcd_index_gather; docodo_amd/csrc/dcd_comm.cpp).
This module is a thin caller: `Comm` wraps a dcd_comm (the 128-byte id travels over whatever rendezvous the host already has --
here torch.distributed, once, at construction), the single-process helpers (`local_merge`, ...) feed the export / merge entry
points directly for the one-GPU tests.  `assign_heavy` is the Python twin of dcd_balance_owners (tests compare them).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check, lib
from .index import Index, _u16

META_BYTES = 32  # sizeof(dcd_term_meta)


class ShardExport:
    """Export buffers of one shard: uint8 tensors on the builder's device + per-owner counts."""

    def __init__(self, term_count, key_count, unit_count, meta, keys, units, max_coord):
        self.term_count, self.key_count, self.unit_count = term_count, key_count, unit_count
        self.meta, self.keys, self.units, self.max_coord = meta, keys, units, max_coord

    def byte_splits(self):
        return (self.term_count * META_BYTES).tolist(), (self.key_count * 2).tolist(), (self.unit_count * 2).tolist()


def heavy_terms(builder, min_units, cap):
    """dcd_build_export_heavy: (fingerprints, units) of the shard's terms with at least min_units posting units."""
    fp, un = np.zeros(max(1, cap), np.uint64), np.zeros(max(1, cap), np.uint64)
    n = C.c_size_t(0)
    fn = lib().dcd_index_export_heavy if isinstance(builder, Index) else lib().dcd_build_export_heavy
    check(fn(builder.h, int(min_units), fp.ctypes.data, un.ctypes.data, cap, C.byref(n)))
    return fp[:n.value].copy(), un[:n.value].copy()


def assign_heavy(lists, totals, n_owners):
    """Owner table of a balanced export.  lists = per rank (fingerprints, units) of its heavy terms, totals = per rank posting
    units.  Heavy terms are taken largest first (ties: smaller fingerprint) and given to the owner with the least heavy units so
    far (ties: lower rank); the LIGHT terms -- everything else, spread by hash -- then fill every owner up to the mean: owner o
    gets the slice of the 32-bit hash space proportional to max(0, mean - heavy_o).  Pure integer function of its arguments:
    every rank computes the same table from the all-gathered lists.  -> (fingerprints, owners, light_thr): fingerprints sorted,
    light_thr[o] = last hash value of owner o's slice (ascending, light_thr[-1] = 2^32 - 1)."""
    units = {}
    for fp, un in lists:
        for f, u in zip(fp.tolist(), un.tolist()):
            units[f] = units.get(f, 0) + u
    total = int(sum(int(t) for t in totals))
    load = [0] * n_owners
    owner = {}
    for f, u in sorted(units.items(), key=lambda x: (-x[1], x[0])):
        o = min(range(n_owners), key=lambda i: (load[i], i))
        owner[f] = o
        load[o] += u
    # light capacity per owner: what is missing to the mean (an owner whose heavy terms already exceed it gets no light term)
    cap = [max(0, total - n_owners * l) for l in load]              # (= n_owners * (mean - heavy_o), kept integral)
    if sum(cap) == 0:
        cap = [1] * n_owners
    thr, cum, tot = [], 0, sum(cap)
    for o in range(n_owners):
        cum += cap[o]
        thr.append(max(0, (cum << 32) // tot - 1))
    thr[-1] = 0xFFFFFFFF
    for o in range(n_owners - 2, -1, -1):                           # an owner without capacity: an empty slice (never below its left neighbour)
        thr[o] = min(thr[o], thr[o + 1])
    fps = sorted(owner)
    return np.array(fps, np.uint64), np.array([owner[f] for f in fps], np.uint32), np.array(thr, np.uint32)


def balance_table(lists, totals, n_owners):
    """dcd_balance_owners: the C twin of assign_heavy (same arguments, same result)."""
    off = np.zeros(len(lists) + 1, np.uint64)
    off[1:] = np.cumsum([len(fp) for fp, _ in lists])
    fp = np.ascontiguousarray(np.concatenate([np.asarray(f, np.uint64) for f, _ in lists]) if lists else np.empty(0, np.uint64))
    un = np.ascontiguousarray(np.concatenate([np.asarray(u, np.uint64) for _, u in lists]) if lists else np.empty(0, np.uint64))
    tot = np.ascontiguousarray(totals, np.uint64)
    cap = max(1, fp.size)
    ofp, oow, thr = np.zeros(cap, np.uint64), np.zeros(cap, np.uint32), np.zeros(n_owners, np.uint32)
    n = C.c_size_t(0)
    check(lib().dcd_balance_owners(n_owners, len(lists), off.ctypes.data, fp.ctypes.data, un.ctypes.data, tot.ctypes.data, ofp.ctypes.data,
                                   oow.ctypes.data, cap, C.byref(n), thr.ctypes.data))
    return ofp[:n.value].copy(), oow[:n.value].copy(), thr


class Comm:
    """A dcd_comm: one communicator per GPU.  transport: _lib.DCD_COMM_RCCL (ncclSend / ncclRecv over xGMI) or _lib.DCD_COMM_HOST
    (buffers staged through shared memory: tests and rehearsals with several ranks on one GPU)."""

    def __init__(self, n_ranks, rank, comm_id, device, transport=_lib.DCD_COMM_RCCL):
        self.n, self.rank, self.device, self.transport = n_ranks, rank, device, transport
        self.h = C.c_void_p()
        idb = (C.c_uint8 * _lib.DCD_COMM_ID_BYTES).from_buffer_copy(bytes(comm_id)) if comm_id is not None else None
        check(lib().dcd_comm_create(n_ranks, rank, idb, device, transport, C.byref(self.h)))

    @staticmethod
    def unique_id(transport=_lib.DCD_COMM_RCCL):
        idb = (C.c_uint8 * _lib.DCD_COMM_ID_BYTES)()
        check(lib().dcd_comm_unique_id(transport, idb))
        return bytes(idb)

    @classmethod
    def from_torch(cls, device, group=None, transport=None):
        """Rendezvous over an initialised torch.distributed group: rank 0 makes the id, one broadcast carries it.  transport
        None: RCCL when the group's backend is nccl, else the host transport."""
        import torch.distributed as dist
        n, rank = dist.get_world_size(group), dist.get_rank(group)
        if transport is None:
            transport = _lib.DCD_COMM_RCCL if dist.get_backend(group) == "nccl" else _lib.DCD_COMM_HOST
        box = [cls.unique_id(transport) if rank == 0 else None]
        if n > 1:
            dist.broadcast_object_list(box, src=0, group=group)
        return cls(n, rank, box[0], device, transport)

    def allgather(self, arr):
        """Small host arrays: -> [n_ranks, ...] (dcd_comm_allgather)."""
        a = np.ascontiguousarray(arr)
        out = np.zeros((self.n,) + a.shape, a.dtype)
        check(lib().dcd_comm_allgather(self.h, a.ctypes.data, a.nbytes, out.ctypes.data))
        return out

    def close(self):
        if getattr(self, "h", None):
            lib().dcd_comm_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def export_shard(builder, n_owners, device=None, overrides=None):
    """dcd_build_export_plan (overrides = (fingerprints, owners[, light_thr]) from assign_heavy: dcd_build_export_plan_balanced) +
    dcd_build_export_fill into torch-owned device buffers."""
    tc = np.zeros(n_owners, np.uint64)
    kc = np.zeros(n_owners, np.uint64)
    uc = np.zeros(n_owners, np.uint64)
    if isinstance(builder, Index):               # (a shard that is itself a merged index: several segments of one rank)
        return export_index(builder, n_owners, device, overrides)
    if overrides is not None:
        ofp, oow = np.ascontiguousarray(overrides[0], np.uint64), np.ascontiguousarray(overrides[1], np.uint32)
        thr = np.ascontiguousarray(overrides[2], np.uint32) if len(overrides) > 2 and overrides[2] is not None else None
        check(lib().dcd_build_export_plan_balanced(builder.h, n_owners, ofp.ctypes.data if ofp.size else None, oow.ctypes.data if ofp.size else None,
                                                   ofp.size, thr.ctypes.data if thr is not None else None, tc.ctypes.data, kc.ctypes.data, uc.ctypes.data))
    else:
        check(lib().dcd_build_export_plan(builder.h, n_owners, tc.ctypes.data, kc.ctypes.data, uc.ctypes.data))
    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    meta = torch.empty(int(tc.sum()) * META_BYTES, dtype=torch.uint8, device=dev)
    keys = torch.empty(int(kc.sum()) * 2, dtype=torch.uint8, device=dev)
    units = torch.empty(int(uc.sum()) * 2, dtype=torch.uint8, device=dev)
    check(lib().dcd_build_export_fill(builder.h, meta.data_ptr() if meta.numel() else None,
                                      keys.data_ptr() if keys.numel() else None, units.data_ptr() if units.numel() else None))
    return ShardExport(tc, kc, uc, meta, keys, units, builder.stats.max_coord)


# (export_shard hands index handles over to export_index)
def export_index(index, n_owners=1, device=None, overrides=None):
    """Export of an index handle (the owned partition after a merge): same buffers as export_shard."""
    tc = np.zeros(n_owners, np.uint64)
    kc = np.zeros(n_owners, np.uint64)
    uc = np.zeros(n_owners, np.uint64)
    if overrides is not None:
        ofp, oow = np.ascontiguousarray(overrides[0], np.uint64), np.ascontiguousarray(overrides[1], np.uint32)
        thr = np.ascontiguousarray(overrides[2], np.uint32) if len(overrides) > 2 and overrides[2] is not None else None
        check(lib().dcd_index_export_plan_balanced(index.h, n_owners, ofp.ctypes.data if ofp.size else None, oow.ctypes.data if ofp.size else None,
                                                   ofp.size, thr.ctypes.data if thr is not None else None, tc.ctypes.data, kc.ctypes.data, uc.ctypes.data))
    else:
        check(lib().dcd_index_export_plan(index.h, n_owners, tc.ctypes.data, kc.ctypes.data, uc.ctypes.data))
    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    meta = torch.empty(int(tc.sum()) * META_BYTES, dtype=torch.uint8, device=dev)
    keys = torch.empty(int(kc.sum()) * 2, dtype=torch.uint8, device=dev)
    units = torch.empty(int(uc.sum()) * 2, dtype=torch.uint8, device=dev)
    check(lib().dcd_index_export_fill(index.h, meta.data_ptr() if meta.numel() else None,
                                      keys.data_ptr() if keys.numel() else None, units.data_ptr() if units.numel() else None))
    return ShardExport(tc, kc, uc, meta, keys, units, index.MaxCoord)


def merge_received(term_count, key_count, unit_count, meta, keys, units, max_coord, device_index=0, two_phase=False):
    """dcd_merge_shards on buffers that hold the sources' contributions back to back (source order);
    two_phase=True goes through dcd_merge_begin / dcd_merge_finish instead (same result)."""
    if two_phase:
        if meta.is_cuda:
            torch.cuda.current_stream(meta.device).synchronize()
        return PendingMerge(term_count, key_count, unit_count, meta, keys, max_coord, device_index).finish(units)
    tc = np.ascontiguousarray(term_count, np.uint64)
    kc = np.ascontiguousarray(key_count, np.uint64)
    uc = np.ascontiguousarray(unit_count, np.uint64)
    if meta.is_cuda:
        # the buffers may have been produced on torch's stream (torch.cat, collectives); the library works on its own
        torch.cuda.current_stream(meta.device).synchronize()
    h = C.c_void_p()
    check(lib().dcd_merge_shards(device_index, tc.size, tc.ctypes.data, kc.ctypes.data, uc.ctypes.data,
                                 meta.data_ptr() if meta.numel() else None, keys.data_ptr() if keys.numel() else None,
                                 units.data_ptr() if units.numel() else None, int(max_coord), C.byref(h)))
    return Index(h, device_index)


class PendingMerge:
    """dcd_merge_begin / dcd_merge_finish: vocabulary, ranks and output layout from the segment table and the keys
    alone, then the unit copy -- the first half overlaps the all-to-all of the posting units (see `distributed_merge`)."""

    def __init__(self, term_count, key_count, unit_count, meta, keys, max_coord, device_index=0):
        tc = np.ascontiguousarray(term_count, np.uint64)
        kc = np.ascontiguousarray(key_count, np.uint64)
        uc = np.ascontiguousarray(unit_count, np.uint64)
        self._keep = (meta, keys)            # the segment table is read again by finish
        self.device_index = device_index
        self.h = C.c_void_p()
        check(lib().dcd_merge_begin(device_index, tc.size, tc.ctypes.data, kc.ctypes.data, uc.ctypes.data,
                                    meta.data_ptr() if meta.numel() else None, keys.data_ptr() if keys.numel() else None,
                                    int(max_coord), C.byref(self.h)))

    def finish(self, units):
        h, self.h = self.h, None
        out = C.c_void_p()
        check(lib().dcd_merge_finish(h, units.data_ptr() if units.numel() else None, C.byref(out)))
        self._keep = None
        return Index(out, self.device_index)

    def __del__(self):
        if getattr(self, "h", None):
            lib().dcd_merge_free(self.h)
            self.h = None


def _split_offsets(counts):
    off = np.zeros(len(counts) + 1, np.int64)
    off[1:] = np.cumsum(np.asarray(counts, np.int64))
    return off


def local_merge(exports, owner, device_index=0, two_phase=False):
    """Single-process stand-in for the all-to-all: gathers, from every shard's export, the slice that
    belongs to `owner` (in shard order) and merges it.  Used by the single-GPU tests: G fake shards on
    one device must reproduce the one-shot build."""
    G = len(exports)
    metas, keyss, unitss = [], [], []
    tc, kc, uc = np.zeros(G, np.uint64), np.zeros(G, np.uint64), np.zeros(G, np.uint64)
    for s, e in enumerate(exports):
        mo, ko, uo = (_split_offsets(x) for x in e.byte_splits())
        metas.append(e.meta[mo[owner]:mo[owner + 1]])
        keyss.append(e.keys[ko[owner]:ko[owner + 1]])
        unitss.append(e.units[uo[owner]:uo[owner + 1]])
        tc[s], kc[s], uc[s] = e.term_count[owner], e.key_count[owner], e.unit_count[owner]
    meta, keys, units = torch.cat(metas), torch.cat(keyss), torch.cat(unitss)
    return merge_received(tc, kc, uc, meta, keys, units, max(e.max_coord for e in exports), device_index, two_phase)


def distributed_merge(builder, comm, balanced=True):
    """dcd_build_merge_distributed / dcd_index_merge_distributed: export -> balanced owners -> all-to-all-v -> merge, inside the
    library.  Returns this rank's owned sub-index.  `builder`: a finished Builder, or an Index handle that holds this rank's shard
    (several segments merged on the device)."""
    out = C.c_void_p()
    fn = lib().dcd_index_merge_distributed if isinstance(builder, Index) else lib().dcd_build_merge_distributed
    check(fn(builder.h, comm.h, 1 if balanced else 0, C.byref(out)))
    return Index(out, comm.device)


def local_gather(owned, device_index=0):
    """Single-process stand-in for gather_index: all owned partitions -> one complete index."""
    exports = [export_index(ix, 1) for ix in owned]
    return local_merge(exports, 0, device_index)


def gather_index(owned, comm, dst=0):
    """dcd_index_gather: the hash partitions -> the complete index on rank `dst` (None elsewhere); its ".index" is what a single
    builder writes for the whole corpus.  (Attach the page table with Index.set_list_image or dist.set_uniform_pages.)"""
    out = C.c_void_p()
    check(lib().dcd_index_gather(owned.h, comm.h, dst, C.byref(out)))
    return Index(out, comm.device) if out.value else None


def allgather_index(owned, comm):
    """dcd_index_allgather: every rank ends with a REPLICA of the complete index (what CreateAsync ends with: one searchable
    index, Index.cs:488, 512 -- here on every GPU, so that query batches shard by query with no routing)."""
    out = C.c_void_p()
    check(lib().dcd_index_allgather(owned.h, comm.h, C.byref(out)))
    return Index(out, comm.device)


def set_uniform_pages(index, n_pages, page_chars, first_doc=0, source="syn"):
    a, p, n = _u16(source)
    check(lib().dcd_index_set_uniform_pages(index.h, n_pages, page_chars, first_doc, p, n))


class SegmentedBuilder:
    """Corpora beyond one build segment (2^32 characters / postings): the reference flushes a temp index every
    MaxTmpIndexItems postings and merges them at the end (Build.cs:335-340, Index.cs:537-689); here every segment is a
    doc-range shard with GLOBAL coordinates, built and exported on the device, and `finish` merges the exports
    (dcd_merge_shards: first delta of every appended segment re-coded) into one index.  Uniform device-resident pages
    (the synthetic corpora); one document per page.  first_page: this builder covers the pages from first_page on (one rank's
    doc range of a distributed build)."""

    def __init__(self, page_chars, max_segment_chars=3_584_000_000, device=0, source="syn", first_page=0, defer_export=False):
        self.page_chars, self.device, self.source = page_chars, device, source
        self.max_pages = max(1, max_segment_chars // page_chars)
        self.first_page = first_page
        self.defer_export = defer_export           # keep the finished builders (finish_distributed exports a single segment directly)
        self.exports, self.builders, self.n_pages, self.n_postings, self.device_ms = [], [], 0, 0, 0.0

    def add_pages_dev(self, text_dev_ptr, n_pages):
        """Pages [first_page + self.n_pages, ... + n_pages) of the corpus, resident at text_dev_ptr."""
        from .index import Builder
        done = 0
        while done < n_pages:
            k = min(self.max_pages, n_pages - done)
            first = self.first_page + self.n_pages + done
            b = Builder(coord_base=first * self.page_chars, device=self.device, shard_only=True)      # (exported and merged below)
            b.AddUniformPagesDev(text_dev_ptr + done * self.page_chars, k, self.page_chars, first_doc=first, source=self.source)
            st = b.Finish()
            self.n_postings += st.n_postings
            self.device_ms += st.ms_total
            if self.defer_export:
                self.builders.append(b)
            else:
                self.exports.append(export_shard(b, 1))
                b.close()
            done += k
        self.n_pages += n_pages

    @property
    def n_segments(self):
        return len(self.exports) + len(self.builders)

    def _export_all(self):
        for b in self.builders:
            self.exports.append(export_shard(b, 1))
            b.close()
        self.builders = []

    def finish(self, pages=True):
        self._export_all()
        ix = local_merge(self.exports, 0, self.device)
        self.exports = []
        if pages:
            set_uniform_pages(ix, self.n_pages, self.page_chars, self.first_page, self.source)
        return ix

    def finish_distributed(self, comm, balanced=True):
        """This rank's doc range -> its hash partition of the global index (export -> all-to-all -> merge).  One segment: the
        builder is exported to the owners directly; several: they are merged on this device first (one shard per rank)."""
        if len(self.builders) == 1 and not self.exports:
            b = self.builders.pop()
            owned = distributed_merge(b, comm, balanced)
            b.close()
            return owned
        part = self.finish(pages=False)
        owned = distributed_merge(part, comm, balanced)
        part.close()
        return owned


class OutOfCoreBuilder:
    """A corpus whose INDEX does not fit the device: the reference's temp-index flush (Build.cs:335-340: a temp index goes to disk
    whenever MaxTmpIndexItems postings have accumulated) and MergeAll (Index.cs:537-573) with the device as the merge engine and
    host memory or files as the tier below it.

      add_pages_dev   every segment (a doc-range shard with global coordinates, shard-only) is built on the device, exported into
                      `partitions` hash partitions and SPILLED -- spill="host": pinned host memory, spill="disk": files
                      <work_dir>/seg<S>.<kind>.npy (the N.tmpind of this path) -- and its device memory released;
      finish(out)     partition by partition: the segments' slices come back to the device, dcd_merge_shards merges them (first
                      delta of every appended segment re-coded), the partition is saved as a sorted run; the runs have disjoint
                      keys and are merged into <out>/.index by a streaming k-way merge of records (dcd_merge_index_files).
                      Peak device memory = one segment build, or one partition of the index -- never the whole index.
                      Returns the index opened ON DISK (Index.LoadOnDisk): its postings stay in the file.

    Uniform device-resident pages (the synthetic corpora); one document per page."""

    def __init__(self, page_chars, work_dir, partitions=8, max_segment_chars=3_584_000_000, device=0, source="syn", spill="disk"):
        import os
        assert spill in ("disk", "host")
        self.page_chars, self.work_dir, self.K, self.device, self.source, self.spill = page_chars, work_dir, partitions, device, source, spill
        self.max_pages = max(1, max_segment_chars // page_chars)
        os.makedirs(work_dir, exist_ok=True)
        self.segments = []                    # per segment: counts [3][K], max_coord, and the three buffers (arrays or file paths)
        self.n_pages, self.n_postings, self.spilled_bytes = 0, 0, 0

    def add_pages_dev(self, text_dev_ptr, n_pages):
        import os
        from .index import Builder
        done = 0
        while done < n_pages:
            k = min(self.max_pages, n_pages - done)
            first = self.n_pages + done
            b = Builder(coord_base=first * self.page_chars, device=self.device, shard_only=True)
            b.AddUniformPagesDev(text_dev_ptr + done * self.page_chars, k, self.page_chars, first_doc=first, source=self.source)
            st = b.Finish()
            self.n_postings += st.n_postings
            e = export_shard(b, self.K)
            b.close()
            seg = {"tc": e.term_count.copy(), "kc": e.key_count.copy(), "uc": e.unit_count.copy(), "max_coord": int(e.max_coord)}
            for kind, t in (("meta", e.meta), ("keys", e.keys), ("units", e.units)):
                if self.spill == "host":
                    h = torch.empty(t.numel(), dtype=torch.uint8, pin_memory=True)
                    h.copy_(t)
                    seg[kind] = h
                else:
                    path = os.path.join(self.work_dir, "seg%d.%s.npy" % (len(self.segments), kind))
                    np.save(path, t.cpu().numpy())
                    seg[kind] = path
                self.spilled_bytes += t.numel()
            del e
            self.segments.append(seg)
            _lib.pool_trim(self.device)       # (the point of the exercise: the device keeps nothing of a finished segment)
            done += k
        self.n_pages += n_pages

    def _slice(self, seg, kind, counts, width, owner):
        off = _split_offsets((counts * width).tolist())
        a, b = int(off[owner]), int(off[owner + 1])
        src = seg[kind]
        if isinstance(src, str):
            arr = np.load(src, mmap_mode="r")[a:b]          # only the partition's bytes are read from the file
            return torch.from_numpy(np.ascontiguousarray(arr))
        return src[a:b]

    def finish(self, out_dir, resident_bytes=0):
        import ctypes as C
        import os
        os.makedirs(out_dir, exist_ok=True)
        dev = torch.device("cuda", self.device)
        runs = []
        S = len(self.segments)
        mx = max([s["max_coord"] for s in self.segments] or [0])
        for k in range(self.K):
            tc = np.array([s["tc"][k] for s in self.segments], np.uint64)
            kc = np.array([s["kc"][k] for s in self.segments], np.uint64)
            uc = np.array([s["uc"][k] for s in self.segments], np.uint64)
            bufs = []
            for kind, name, width in (("meta", "tc", META_BYTES), ("keys", "kc", 2), ("units", "uc", 2)):
                parts = [self._slice(s, kind, s[name], width, k) for s in self.segments]
                bufs.append(torch.cat(parts).to(dev) if parts else torch.empty(0, dtype=torch.uint8, device=dev))
            ix = merge_received(tc, kc, uc, bufs[0], bufs[1], bufs[2], mx, self.device)
            del bufs
            set_uniform_pages(ix, self.n_pages, self.page_chars, 0, self.source)
            run_dir = os.path.join(self.work_dir, "run%d" % k)
            os.makedirs(run_dir, exist_ok=True)
            ix.Save(run_dir)                                  # sorted run of this partition (+ the page list, the same for every run)
            ix.close()
            _lib.pool_trim(self.device)
            runs.append(os.path.join(run_dir, ".index").encode("utf-8"))
        arr = (C.c_char_p * len(runs))(*runs)
        check(lib().dcd_merge_index_files(arr, len(runs), os.path.join(out_dir, ".index").encode("utf-8")))
        import shutil
        shutil.copyfile(os.path.join(self.work_dir, "run0", ".index.list"), os.path.join(out_dir, ".index.list"))
        self.segments = []
        return Index.LoadOnDisk(out_dir, self.device, resident_bytes)
