"""Device-side classification driver (host half).

Replaces the reference's ``workflow.assign_readmap`` (woltka/workflow.py:
941-1058) and the assigners / counters of ``woltka/classify.py``: an ``Engine``
owns one GPU context, the flattened hierarchy and one *job* per requested rank;
``run_chunk`` packs a chunk, runs all ranks in one kernel pass and
``collect`` folds the exact integer counts back into the reference's
``data[rank][sample]`` dicts.

Count keys carry a *group* = index of a (sample, stratum) pair, so one device
table serves demultiplexed and stratified runs alike.
"""
import os

import numpy as np

from . import _native as nat
from .align import pack_queries
from .hierarchy import FeatureIndex, flatten_hierarchy
from .hostio import (MAX_GROUPS, ROUTES, MapWriter, StageRing,  # noqa: F401
                     _NO_TREE_ROOT, _BySubject, _Staged, _prefetch,
                     _take_context, cpu_budget, drop_context_ahead,
                     open_context_ahead, take_warm_tokenizer,
                     tokenizer_threads, warm_tokenizer_ahead)
from .routes.coords import CoordMatchRoute
from .routes.device_text import DeviceTextRoute
from .routes.fold import Folding, exact_to_numbers  # noqa: F401
from .routes.readmaps import ReadMaps
from .routes.replay import Replay
from .routes.words import WordsRoute


class Engine(DeviceTextRoute, WordsRoute, CoordMatchRoute, ReadMaps, Replay,
             Folding):
    """One classification job on one GPU.

    Parameters mirror the assignment options of ``workflow.classify``
    (workflow.py:162-187).  ``major`` is the percentage given on the command
    line; like the reference it is turned into the fraction ``major / 100``
    (workflow.py:276) and compared in binary64 on the device.
    """

    def __init__(self, tree, rankdic, root, ranks, uniq=False, major=None,
                 above=False, subok=False, unasgd=False, device=0,
                 table_slots=None, sizes=None, major_frac=None):
        self.ctx = _take_context(device)
        self.ranks = list(ranks)
        self.use_tree = bool(tree)
        native = getattr(tree, 'native', None)
        if tree and native is not None and (
                rankdic is None or getattr(rankdic, 'native', None) is native):
            # the dicts are views of a natively built table
            # (workflow.build_hierarchy): its pre-order arrays as they are
            self.hier = native.hierarchy()
        elif tree:
            self.hier = flatten_hierarchy(
                dict(tree) if native is not None else tree,
                dict(rankdic) if getattr(rankdic, 'native', None) is not None
                else rankdic, root)
        else:
            # `free` on an empty hierarchy still has defined results
            # (classify.py:73-78): keep a lone root so that every subject is
            # "not in the tree"
            self.hier = flatten_hierarchy({_NO_TREE_ROOT: _NO_TREE_ROOT})
        self.index = self.hier.index
        h = self.hier
        self.ctx.set_tree(h.parent, h.last, h.rank_code)
        flags = 0
        if uniq:
            flags |= nat.F_UNIQ
        if above:
            flags |= nat.F_ABOVE
        if subok:
            flags |= nat.F_SUBOK
        if unasgd:
            flags |= nat.F_UNASSIGNED
        # --sizes: every contribution is logged as (feature, subject, divisor)
        # and weighted by sizes[subject] on the host (classify.py:174-297)
        self.sizes = sizes
        self.sized = {}                 # (job, group key, feature, subj, div) -> n
        if sizes:
            flags |= nat.F_SIZED
            self.ctx.log_reserve(1 << 22)
        self.jobs, self.modes, self.slots = [], [], []
        slot_of = {}
        for rank in self.ranks:
            if rank is None or rank == 'none' or tree is None:
                # workflow.py:1017: also taken when no hierarchy object exists
                mode, slot, frac = nat.MODE_NONE, 0, 0.0
            elif rank == 'free':
                mode, slot, frac = nat.MODE_FREE, 0, 0.0
            else:
                mode = nat.MODE_RANK
                code = h.code_of(rank)
                if code not in slot_of:
                    if len(slot_of) >= nat.MAX_RANK_SLOTS:
                        raise ValueError('Too many distinct ranks.')
                    slot_of[code] = len(slot_of)
                    self.ctx.build_rank_table(slot_of[code], code)
                slot = slot_of[code]
                # (assign_readmap's callers hand over the fraction itself)
                frac = major_frac if major_frac else \
                    (major / 100) if major else 0.0
            self.jobs.append(nat.Job(mode, slot, flags, 0, frac))
            self.modes.append(mode)
            self.slots.append(slot)
        self._anc = {}                      # slot -> downloaded rank table
        self.groups = []                    # group id -> (sample, stratum)
        self.group_ids = {}
        # the count table grows with what the run needs (`_ensure_table`); it
        # is never left to fill up: a full table loses counts
        self.slots_reserved = 0
        self._table_fixed = table_slots
        self._reserve(table_slots or max(1 << 20, 4 * len(self.index)))
        self._job_base = 0                  # first job of the batch in flight
        self._final = {}                    # (rank, sample) -> (units, big) of the last `finish`
        # large folds of the count table stay arrays (cells.py): (job, sample
        # index, stratum index, feature, units) per fold, until `finish`
        self._stash, self._stash_big = [], []
        self._lz_samples, self._lz_sample_ids = [], {}
        self._lz_strata, self._lz_strata_ids = [], {}
        self._n_reads = 0                   # reads classified (bounds the mapper chunks)
        self._n_files = 0                   # alignment files begun (each restarts the mapper's chunks)
        self._replay = None
        self._writer = None                 # MapWriter of the native read maps
        self._map_pool, self._map_seq, self._map_jobs = None, None, []   # their formatting threads
        self._subj_feat_arr = None
        self._read_pool = None
        self._strata_ahead = None   # (thread, box) of a strata map being read ahead
        self._dtok_lap = {}     # (WOLTKA_DTOK_TIMING: seconds inside _run_dtok)
        self.genes = None
        self.gene_feature = None
        self._pairs_by_index = False
        # dense subject indices (order of first appearance in the alignments)
        # -> feature ids, mirrored on the device by wk_set_subjects
        self.subjects = FeatureIndex()
        self.subj_feature = []
        # native SAM tokenizer (created on first use) and the translation of
        # its subject ids into `self.subjects` indices / genome indices
        self.tok = None
        self._tok_samples = []          # sample names of the native demultiplexer
        self._smap_key, self._smap = None, None
        self._exclude = None
        self._gmap_key, self._gmap = None, None
        self._epoch = 0
        self._used_bound, self._used_epoch = (0, None), -1
        self._units, self._big = {}, {}     # folded counts until `finish`
        self._tok_map = np.empty(0, dtype=np.int32)
        self._tok_identity = True
        self._tok_map_sent = 0      # entries of _tok_map the device holds (wk_dtok_subject_map)
        self._dtrimsub = None       # `--trim-sub` of the file on the device text route
        self._dexclude = None       # ... and its `--exclude` set
        self._tok_genome = np.empty(0, dtype=np.int32)
        self._tok_cover = np.empty(0, dtype=np.int64)
        self._ring, self._ring_prev = None, None    # packed-record staging
        self._oring = None                          # coord-match staging
        self._tring, self._reader = None, None      # device tokenizer: text staging, reader threads
        self._fused_seen = (0, 0)                   # wk_dtok_fused_counts when the last file ended
        self._deferred_from = None                  # see take_deferred
        self._hits_open = None                      # group of the hits piled up on the device (`_settle_hits`)
        # read maps formatted on the device (csrc/wk_readmap.hpp)
        self._dmaps = None          # (rank2dir, outzip, namedic) while a file is read that way
        self._dfmt = 'sam'          # format of the file the device tokenises
        self._dpath = None          # ... and its path
        self._dmaps_n = -1          # subjects the device's read-map tables cover
        self._dmaps_ok = False
        self._mring = None          # pinned buffers the map text is fetched into
        # strata map joined on the device (csrc/wk_strata.hpp)
        self._dstrata = None        # {'fp', 'labels', 'slots', 'key'} while the device holds the sample's map
        self._sbuf = [None, None]   # pinned buffers the map text is inflated into
        self._sbuf_next = 0

    def close_later(self):
        """`close` on a thread of its own: giving the device buffers and the
        pinned rings back takes ~0.07 s that the caller can spend rounding and
        writing the profiles.  (Not a daemon: the interpreter waits for it.)"""
        import threading
        if self._map_pool is not None:      # (errors of the map writers: here)
            self._maps_done()
        threading.Thread(target=self.close, name='wk-close').start()

    def close(self):
        try:
            if self._map_pool is not None:
                try:
                    self._maps_done()
                finally:
                    self._map_pool.shutdown(wait=True)
                    self._map_seq.shutdown(wait=True)
                    self._map_pool = self._map_seq = None
            if self._writer is not None:
                self._writer.close()
                self._writer = None
        finally:
            if self._strata_ahead is not None:
                self._strata_ahead[0].join()
                self._strata_ahead = None
            # (a reader started ahead that no file of this run took over: it
            # copies into this context)
            from .routes.device_text import drop_text_ahead
            drop_text_ahead()
            if self._read_pool is not None:
                self._read_pool.shutdown(wait=True)
                self._read_pool = None
            if self.tok is not None:
                self.tok.close()
            if self._reader is not None:
                self._reader.close()
            self.ctx.close()

    # ------------------------------------------------------------------
    MAX_SLOTS = 1 << 30
    GROW_TO = 1 << 26       # slots (16 B each) a table grows to by itself when it keeps filling up

    def _reserve(self, slots):
        """(Re)allocate the device count table (cleared) with a power-of-two
        number of slots >= ``slots``."""
        n = 1024
        while n < min(int(slots), self.MAX_SLOTS):
            n <<= 1
        self.ctx.counts_reserve(n)
        self.slots_reserved = n

    def _ensure_table(self, data, n_records, n_groups):
        """Room for the keys the next chunk can add: at most one per record and
        job, and at most one per (group of the chunk, feature, job).  Counts
        are exact integers, so they can be folded to the host at any point
        (`collect`) — done here when the table is more than a quarter full —
        and the table re-allocated larger when one chunk needs it.  (The
        reference's dicts have no size limit; a fixed table used to fail at the
        very end of a long multi-sample run.)"""
        n_jobs = min(len(self.jobs), nat.MAX_JOBS)
        need = min(n_records + 1, max(1, n_groups) * (len(self.index) + 1)) \
            * n_jobs
        # every key the groups met so far (and this chunk's) could ever hold:
        # when even that fits with room to spare the table need not be asked
        # (asking waits for the device)
        most = (len(self.groups) + max(1, n_groups)) * (len(self.index) + 2) \
            * n_jobs
        if 2 * most <= self.slots_reserved:
            # (keys added from here on are in no bound: the table is asked
            # again before a bound is trusted)
            self._used_bound = (0, None)
            return
        # (asking counts the table's keys on the device and waits: 0.5 ms per
        # block of a stratified run.  The keys in use are at most those counted
        # last plus what the chunks since then could have added: only when that
        # bound is too large is the table asked again)
        known, table = getattr(self, '_used_bound', (0, None))
        if table != self.slots_reserved or self._used_epoch != self._epoch:
            known = None
        if known is not None and 2 * (known + need) <= self.slots_reserved \
                and 4 * known <= self.slots_reserved:
            self._used_bound = (known + need, self.slots_reserved)
            return
        self._settle_hits()     # (their keys are in the bound, not in the table yet)
        used = self.ctx.stats()['table_used']
        self._used_epoch = self._epoch
        if 2 * (used + need) <= self.slots_reserved and \
                4 * used <= self.slots_reserved:
            self._used_bound = (used + need, self.slots_reserved)
            return
        self.collect(data)
        self._used_bound = (need, None)     # (asked again next time)
        if 2 * need > self.slots_reserved and not self._table_fixed:
            self._reserve(4 * need)
        elif not self._table_fixed and self.slots_reserved < self.GROW_TO:
            # a table that had to be folded because it filled up comes back
            # four times the size (HBM is what this device has to spare): a
            # stratified run folds once or twice instead of once per few
            # blocks, and `finish` adds up that many copies of a cell less
            self._reserve(min(4 * self.slots_reserved, self.GROW_TO))

    def load_strata(self, fp, zippers, then=None, device=False):
        """Read-to-stratum map of one sample into the native tokenizer;
        returns the stratum labels (workflow.read_strata, workflow.py:912-938).
        ``then``: the map that will be asked for next — read on a thread into
        the tokenizer's second table while this sample is tokenised.
        ``device``: the alignments of this sample will be tokenised on the
        device: the join table is built there (csrc/wk_strata.hpp)."""
        from os.path import basename
        if self.tok is None:
            self.tok = nat.Tokenizer(tokenizer_threads(), self._exclude)
        self._dstrata = None
        if device and not os.environ.get('WOLTKA_NO_DSTRATA'):
            labels = self._load_strata_device(fp, zippers, then)
            if labels is not None:
                return labels
        else:
            self.ctx.strata_clear()
        labels = None
        ahead, self._strata_ahead = self._strata_ahead, None
        if ahead is not None:
            thread, box = ahead
            thread.join()
            if box['fp'] == fp and 'text' not in box and (
                    'labels' in box or 'err' in box):
                if 'err' in box:
                    raise box['err']
                self.tok.strata_swap()
                labels = box['labels']
        if labels is None:
            labels = self._read_strata(fp, zippers, False)
        if then is not None and then != fp:
            import threading
            box = {'fp': then}

            def work():
                try:
                    box['labels'] = self._read_strata(then, zippers, True)
                except Exception as e:      # raised when the map is asked for
                    box['err'] = e
            thread = threading.Thread(target=work, name='wk-strata')
            self._strata_ahead = (thread, box)
            thread.start()
        if not labels:
            raise ValueError('No stratification information is found in file: '
                             f'{basename(fp)}.')
        return labels

    def _read_strata(self, fp, zippers, ahead):
        from . import pgzip
        from .file import readzip_bytes
        fh = pgzip.open_parallel(fp) if fp.endswith('.gz') else None
        with (fh if fh is not None else readzip_bytes(fp, zippers)) as fh:
            return self.tok.load_strata(fh, ahead=ahead)

    def native_chunks(self, stream, head, exclude, block_bytes, ordinal,
                      want_names, trimsub=None, want_groups=False,
                      want_strings=True, want_samples=False, cover=None,
                      fmt='sam', part=None, words=False, dmaps=None,
                      keep_empty=False, words_dev=False):
        """SAM text -> packed chunks through the native tokenizer.  Yields
        (reads or None, packed, strata ids, name descriptors, sample ids,
        ranges) where packed = (subj, qoff) of subject indices, or for
        coord-match (genome, beg, end, length, hoff).  With ``cover`` (a
        ``ranges.Coverage``) the "ex" columns are produced for plain
        classification too and ``ranges`` = (coverage subject id, beg, end) per
        record.  ``keep_empty``: hits of aligned length 0 stay in the arrays
        (`regroup_hits` counts them where ordinal.py:222 does, then drops
        them)."""
        from .align import native_sam_blocks
        if self.tok is None:
            # (a tokenizer whose dictionary was filled from this file's first
            # bytes while the hierarchy was read: hostio.warm_tokenizer_ahead)
            warm = take_warm_tokenizer(getattr(stream, 'name', None))
            if warm is not None and exclude:
                warm.close()
                warm = None
            self.tok = warm or nat.Tokenizer(tokenizer_threads(), exclude)
        tok = self.tok
        device_ex = ordinal and cover is None and not want_names and \
            (not want_groups or self._dstrata is not None) and \
            not want_samples and len(self.jobs) <= nat.MAX_JOBS
        # (the device tokenises SAM, BLAST tabular text and PAF in both
        # flavours, simple maps in the plain one -- they have no other:
        # align.py:258-547, 621-674, 753-981, 984-1213)
        # (`words_dev`: plain assigners whose packed records only the device
        # text route can make -- `--trim-sub`, where the tokenizer's names
        # are not the subjects and the kernels translate)
        # (`--exclude`: the plain flavour's kernels drop the runs that hit a
        # name of the set; not with read maps, not the "ex" parsers' way)
        if (words or words_dev or device_ex or dmaps) and (
                fmt in ('sam', 'b6o', 'paf') or (fmt == 'map' and
                                                 not ordinal)) and \
                (not exclude or ((words or words_dev) and not dmaps)) and \
                not os.environ.get('WOLTKA_NO_DTOK'):
            from .align import _parallel_reader, part_range
            from .file import GunzipStream
            reader = _parallel_reader(stream, tok, None)
            if part is not None:
                # one of several byte ranges of a large plain file (`--gpus N`
                # on a single file, shard.FilePart): the range's text through
                # the device like a file that begins and ends there
                reader = part_range(reader, part, fmt, bool(ordinal)) \
                    if reader is not None else None
            if reader is None and isinstance(stream, GunzipStream):
                # a gzip file inflated by this package's own decoder: its
                # blocks go from the inflater's threads straight into the
                # pinned buffers the device copies from (a line read to
                # tell the format goes back in front)
                stream.unread(head)
                head = b''
                reader = stream
            if reader is not None:
                # the text goes to the GPU as it is: tokenised there (with
                # `dmaps` the read maps are formatted there too)
                self._dmaps = dmaps if not (words or device_ex) else None
                self.ctx.dtok_keep_reads(self._dmaps is not None)
                self._dfmt = fmt
                self._dtrimsub = trimsub if not ordinal else None
                self._dexclude = set(exclude) if exclude else None
                if exclude:
                    # (the tokenizer's ids are no subject indices from now
                    # on: names of the set get none; the device is told
                    # before its first scan that a map is coming)
                    self._tok_identity = False
                    if self._tok_map_sent != self._tok_map.size or \
                            not self._tok_map.size:
                        self.ctx.dtok_subject_map(self._tok_map)
                        self._tok_map_sent = self._tok_map.size
                self._dpath = getattr(stream, 'name', None)
                self.ctx.dtok_format(fmt)
                try:
                    yield from self._device_chunks(reader, block_bytes,
                                                   ordinal=bool(ordinal))
                finally:
                    self._dmaps = None
                    if getattr(self.ctx, '_h', None):   # (still open)
                        self.ctx.dtok_keep_reads(False)
                return
        from .routes.device_text import drop_text_ahead
        drop_text_ahead()       # (a reader started for the device route)
        if want_groups and self._dstrata is not None:
            # (the device holds the join table but this file is tokenised on
            # the host after all)
            self._host_strata_table()
        ring = None
        if words:
            if self._ring is None:
                # records of a block: a line has at least ~24 bytes
                self._ring = StageRing(self.ctx, 5, {
                    'packed': (np.uint32, block_bytes // 24 + 4096)})
            ring = self._ring
        elif ordinal and cover is None and not want_names and \
                not want_groups and not want_samples and \
                not os.environ.get('WOLTKA_NO_PINNED'):
            if self._oring is None:
                n = block_bytes // 24 + 4096
                self._oring = StageRing(self.ctx, 4, {
                    'subj': (np.int32, n), 'beg': (np.int32, n),
                    'end': (np.int32, n), 'len': (np.uint32, n),
                    'off': (np.int32, n + 1)})
            ring = self._oring
        state = {'fresh': []}

        def sink(tok_, n_reads, n_records):
            """Between tokenising and fetching a block: the dictionary growth
            belongs to this block (fetched before the tokenizer moves on); the
            coord-match has the tokenizer translate subject ids into genome
            indices of the gene tables on the way out."""
            state['fresh'] = fresh = tok_.new_subjects()
            if ordinal:
                if fresh:
                    gidx = self.genes.genome_index.get
                    self._tok_genome = np.concatenate([
                        self._tok_genome,
                        np.fromiter((gidx(x, -1) for x in fresh), np.int32,
                                    len(fresh))])
                    tok_.set_subject_map(self._tok_genome)
                elif not state.get('mapped'):
                    tok_.set_subject_map(self._tok_genome)
                state['mapped'] = True
            return ring.current() if ring is not None else None

        def blocks():
            if not ordinal:
                tok.set_subject_map(None)
            for buf, res in native_sam_blocks(stream, tok, block_bytes,
                                              extra=3 if (cover is not None
                                                          or keep_empty)
                                              else int(ordinal),
                                              want_names=want_names,
                                              head=head,
                                              want_groups=want_groups,
                                              want_samples=want_samples,
                                              fmt=fmt, part=part,
                                              exclude=exclude, sink=sink):
                if res.get('sunk'):
                    res['slot'] = ring.take()
                fresh, state['fresh'] = state['fresh'], []
                yield buf, res, fresh, \
                    (tok.new_samples() if want_samples else [])

        for buf, res, fresh, fresh_samples in _prefetch(blocks()):
            self._tok_samples.extend(fresh_samples)
            if fresh and cover is not None:
                self._tok_cover = np.concatenate([
                    self._tok_cover,
                    np.fromiter(map(cover.subject, fresh), np.int64,
                                len(fresh))])
            if fresh:
                if ordinal:
                    pass        # (translated by the tokenizer: `sink` above)
                else:
                    base = self._tok_map.size
                    intern = self.subjects.intern
                    if trimsub:
                        fresh = [x.rsplit(trimsub, 1)[0] for x in fresh]
                    ids = np.fromiter((intern(x) for x in fresh), np.int32,
                                      len(fresh))
                    if self._tok_identity and not np.array_equal(
                            ids, np.arange(base, base + ids.size)):
                        self._tok_identity = False
                    self._tok_map = np.concatenate([self._tok_map, ids])
            if 'words' in res:
                del buf
                if res['n_reads']:
                    yield None, ('words', res['words'], res['n_reads'],
                                 res['slot']), None, None, None, None
                else:
                    ring.release(res['slot'])
                continue
            reads = nat.Tokenizer.query_names(buf, res['qname']) \
                if (want_names and want_strings) else None
            names = (buf, res['qname']) if (want_names and not want_strings) \
                else None
            del buf
            if ordinal:
                packed = (res['subj'], res['beg'], res['end'], res['len'],
                          res['off'])
                if 'slot' in res:       # pinned buffers: given back once staged
                    packed = _Staged(packed)
                    packed.slot = res['slot']
            else:
                subj = res['subj'] if self._tok_identity \
                    else self._tok_map[res['subj']]
                packed = (subj, res['off'])
            ranges = None if cover is None else (
                self._tok_cover[res['subj']], res['beg'], res['end'])
            if res['off'].size > 1:
                group = res.get('group')
                if group is not None and self._dstrata is not None:
                    group = self._host_strata_ids(group)
                yield reads, packed, group, names, res.get('sample'), ranges
            elif 'slot' in res:
                ring.release(res['slot'])

    # ------------------------------------------------------------------
    def _group_array(self, n, sample_of, strata_of):
        """Per-read group ids (int32 array), or one id (int) when the whole
        chunk belongs to one (sample, no stratum) pair."""
        gid = self.group_ids
        groups = self.groups

        def get(sample, stratum):
            key = (sample, stratum)
            g = gid.get(key)
            if g is None:
                g = len(groups)
                if g >= MAX_GROUPS:
                    raise ValueError('Too many (sample, stratum) groups in '
                                     'one pass.')
                gid[key] = g
                groups.append(key)
            return g
        per_read_sample = isinstance(sample_of, list)
        if strata_of is None and not per_read_sample:
            # one sample for the whole chunk: one group id (WK_GROUP_UNIFORM)
            return get(sample_of, None)
        out = np.empty(n, dtype=np.int32)
        for i in range(n):
            s = sample_of[i] if per_read_sample else sample_of
            if s is False:
                out[i] = -1
                continue
            if strata_of is None:
                out[i] = get(s, None)
            else:
                t = strata_of[i]
                out[i] = -1 if t is None else get(s, t)
        return out

    def _strata_groups(self, sample, labels, ids):
        """Stratum ids of the native tokenizer -> group ids of (sample, label)
        pairs; -1 (read not in the strata map) stays -1."""
        key = (sample, self._epoch, len(labels))
        if self._gmap_key != key:
            gid, groups = self.group_ids, self.groups
            gmap = np.empty(len(labels), dtype=np.int32)
            for i, lab in enumerate(labels):
                k = (sample, lab)
                g = gid.get(k)
                if g is None:
                    g = len(groups)
                    if g >= MAX_GROUPS:
                        raise ValueError('Too many (sample, stratum) groups '
                                         'in one pass.')
                    gid[k] = g
                    groups.append(k)
                gmap[i] = g
            self._gmap_key, self._gmap = key, gmap
        return np.where(ids >= 0, self._gmap[np.maximum(ids, 0)],
                        -1).astype(np.int32)

    def _sample_groups(self, ids, allow):
        """Sample ids of the native demultiplexer -> group ids of (sample,
        None); samples outside the whitelist -> -1.  Returns (groups, samples
        met)."""
        names = self._tok_samples
        key = (self._epoch, len(names))
        if self._smap_key != key:
            gid, groups = self.group_ids, self.groups
            smap = np.empty(len(names), dtype=np.int32)
            for i, name in enumerate(names):
                if allow is not None and name not in allow:
                    smap[i] = -1
                    continue
                k = (name, None)
                g = gid.get(k)
                if g is None:
                    g = len(groups)
                    gid[k] = g
                    groups.append(k)
                smap[i] = g
            self._smap_key, self._smap = key, smap
        group = self._smap[ids]
        met = {names[i] for i in np.unique(ids).tolist()
               if self._smap[i] >= 0}
        return group, met

    def run_chunk(self, data, reads, subque, sample_of, strata_of, trimsub,
                  rank2dir, outzip, namedic, ordinal, packed=None,
                  strata_ids=None, strata_labels=None, names=None,
                  sample_ids=None, allow=None, packed_is_set=False):
        """Classify one chunk at every rank; returns the number of queries the
        reference would report for it (workflow.py:305).  The dispatcher: a
        chunk that is a block of text on the device or a buffer of packed
        records names its route (``ROUTE_OF``); everything else — chunks of
        the Python parsers, of the host tokenizer, anything that needs the
        assignments back on the host — takes the general route."""
        if packed is not None and isinstance(packed[0], str):
            if packed[0] != 'dhits':
                self._settle_hits()
            return getattr(self, self.ROUTE_OF[packed[0]])(data, packed,
                                                          sample_of)
        self._settle_hits()
        return self._run_general(
            data, reads, subque, sample_of, strata_of, trimsub, rank2dir,
            outzip, namedic, ordinal, packed, strata_ids, strata_labels, names,
            sample_ids, allow, packed_is_set)

    # route of a chunk by the tag of its `packed` tuple
    ROUTE_OF = {'dtok': '_run_dtok',        # text scanned on the device -> packed records (routes/device_text.py)
                'dhits': '_run_dhits',      # text scanned on the device -> coord-match hits
                'words': '_run_words'}      # packed records of the host tokenizer (routes/words.py)

    def _run_general(self, data, reads, subque, sample_of, strata_of, trimsub,
                     rank2dir, outzip, namedic, ordinal, packed, strata_ids,
                     strata_labels, names, sample_ids, allow, packed_is_set):
        """The general route: subject lists (or staged hits) to the device,
        every job through the generic evaluator, assignments back when read
        maps or the replay want them (``packed``: arrays of the native
        tokenizer instead of ``subque``)."""
        n = len(reads) if packed is None else packed[-1].size - 1
        # room for the (sample, stratum) groups this chunk can add
        if sample_ids is not None:
            fresh = len(self._tok_samples) + 1
        elif strata_ids is not None:
            fresh = len(strata_labels) + 1
        elif isinstance(sample_of, list) or strata_of is not None:
            fresh = n + 1
        else:
            fresh = 1
        if len(self.groups) + fresh >= MAX_GROUPS // 2:
            self.collect(data)
        # ... and for its count keys (before the chunk's group ids are taken:
        # folding the table to the host starts a new set of groups)
        if ordinal:     # a hit matches a few genes at most
            n_rec = 4 * int((self._hits if packed is None else packed)[-1][-1])
        elif packed is not None:
            n_rec = int(packed[0].size)
        else:
            n_rec = sum(map(len, subque))
        self._ensure_table(data, n_rec, min(max(n, 1), fresh))
        seen = None
        if sample_ids is not None:
            group, seen = self._sample_groups(sample_ids, allow)
        elif strata_ids is not None:
            group = self._strata_groups(sample_of, strata_labels, strata_ids)
        else:
            group = self._group_array(n, sample_of, strata_of)
        # every sample met in a chunk gets a (possibly empty) profile at every
        # rank, like `data[rank].setdefault(sample, {})` in workflow.py:1058
        if seen is None:
            seen = (set(sample_of) - {False}) \
                if isinstance(sample_of, list) \
                else ({sample_of} if n else set())
        for rank in self.ranks:
            for s in seen:
                data[rank].setdefault(s, {})
        want = rank2dir is not None or self._replay is not None
        map_order = None
        self._n_reads += n
        if ordinal:
            genome, beg, end, length, hoff = \
                self._hits if packed is None else packed
            if np.ndim(group) == 0 and not want and \
                    len(self.jobs) <= nat.MAX_JOBS:
                # one sample, no read maps: match + count in one pass over the
                # reads (wk_ordinal_count), no gene lists.  How many queries
                # matched a gene ("Number of sequences classified",
                # workflow.py:305,344) is asked once per file
                # (`take_deferred`), not per chunk: the answer waits for the
                # device.
                if self._deferred_from is None:
                    self._deferred_from = self.ctx.stats()['n_reads']
                self.ctx.ordinal_stage(genome, beg, end, length, hoff,
                                       self._th)
                self._release_staged(packed)
                self.ctx.set_uniform_group(group)
                self.ctx.ordinal_count(self.jobs)
                assign, nq = None, 0
            else:
                before = self.ctx.stats()['n_reads']
                if np.ndim(group) == 0:  # (the coord-match stage takes an array)
                    group = np.full(hoff.size - 1, group, dtype=np.int32)
                self.ctx.ordinal_stage(genome, beg, end, length, hoff,
                                       self._th, group=group)
                self._release_staged(packed)
                self.ctx.ordinal_match()
                assign = self._classify_staged(data, want)
                nq = (self.ctx.stats()['n_reads'] - before) \
                    // self._n_batches()
            if want:
                subj, qoff = self.ctx.chunk_download()
                if rank2dir is not None and self._replay is None and \
                        reads is not None and names is None:
                    # the reference lists a chunk's queries in the order its
                    # matcher met them, not in input order
                    map_order = self._mapper_order(
                        (genome, beg, end, length, hoff), subj,
                        self.ctx.ordinal_hit_offsets(genome.size))
        else:
            if packed is None:
                subj, qoff = pack_queries(subque, self.subjects, trimsub)
            else:
                subj, qoff = packed
            known = len(self.subj_feature)
            if len(self.subjects) > known:
                self.subj_feature.extend(self.index.intern_many(
                    self.subjects.names[known:]))
                self.ctx.set_subjects(self.subj_feature)
            # reads of more candidates than the count keys can say (k <= 4095)
            # are evaluated here, with exact rationals (classify.counter takes
            # any k, classify.py:156-171), and leave the chunk as empty reads
            if qoff.size > 1 and not want and not self.sizes and \
                    subj.size - n >= nat.MAX_K and \
                    int(np.diff(qoff).max()) > nat.MAX_K:
                subj, qoff = self._fold_huge_reads(subj, qoff, group)
            # the Python parsers and the native tokenizer hand over sets;
            # trimming can merge subjects
            self.ctx.chunk_stage(
                subj, qoff, group=group,
                subj_is_set=(packed_is_set if packed is not None
                             else not trimsub), indexed=True)
            assign = self._classify_staged(data, want)
            if want and (names is None or self._replay is not None):
                subj = self._subject_features()[subj]   # read maps work on feature ids
            elif want:                  # (the native formatter's thread does it)
                subj = _BySubject(subj, self._subject_features())
            nq = n
        if self.sizes:
            self._collect_log()
        if self._replay is not None:
            self._replay_chunk(assign, subj, qoff, group, n)
            return nq
        if want and names is not None:
            # formatted on helper threads (numpy and the native formatter run
            # without the GIL), several chunks at a time, while this thread
            # stages the next chunk; one more thread appends the texts in
            # chunk order.  Arrays that borrow memory (staging slots, a
            # mapped file) are copied first: their owners move on.
            if self._map_pool is None:
                from concurrent.futures import ThreadPoolExecutor
                self._map_pool = ThreadPoolExecutor(max_workers=self.MAP_THREADS)
                self._map_seq = ThreadPoolExecutor(max_workers=1)
            self._maps_done(keep=self.MAP_THREADS)      # (bounds what is held)
            for j, mode in enumerate(self.modes):   # (device calls stay on this thread)
                if mode == nat.MODE_RANK:
                    self._rank_table(self.slots[j])

            def own(a):
                if isinstance(a, _BySubject):
                    return a
                if isinstance(a, memoryview) and \
                        isinstance(a.obj, np.ndarray) and a.obj.flags.owndata:
                    return a        # (a view of a block's own buffer keeps it alive)
                return a if isinstance(a, np.ndarray) and a.flags.owndata \
                    and not isinstance(a, _Staged) else np.array(a)
            job = self._map_pool.submit(
                self._format_maps_native, [own(a) for a in assign], own(subj),
                own(qoff), tuple(own(a) for a in names), sample_of, rank2dir,
                outzip, namedic)
            self._map_jobs.append(self._map_seq.submit(self._append_maps, job))
        elif want:
            self._maps_done()
            self._write_maps(assign, subj, qoff, reads, sample_of, rank2dir,
                             outzip, namedic, order=map_order)
        return nq

    def _subject_features(self):
        """`subj_feature` as an array (kept until the list grows)."""
        if self._subj_feat_arr is None or \
                self._subj_feat_arr.size != len(self.subj_feature):
            self._subj_feat_arr = np.asarray(self.subj_feature, dtype=np.int32)
        return self._subj_feat_arr

    def take_deferred(self):
        """Queries classified since the last call that `run_chunk` has not
        reported yet (the coord-match tally route counts them on the
        device)."""
        self._settle_hits()
        if self._deferred_from is None:
            return 0
        n = (self.ctx.stats()['n_reads'] - self._deferred_from) \
            // self._n_batches()
        self._deferred_from = None
        return n

    def _settle_hits(self):
        """Blocks of the device text route whose hits are piled up on the
        device (`wk_dtok_stage_hits_append`: the match sorted by genome stripe
        wants a few million hits, a block brings 1.5 M) are matched and
        counted now.  Called in front of everything that looks at the counts
        or stages another chunk."""
        group = self._hits_open
        if group is None:
            return
        self._hits_open = None
        self.ctx.set_uniform_group(group)
        self.ctx.ordinal_count(self.jobs)

    def _release_staged(self, packed):
        """The staging call has copied the block (it waits for its copies):
        its pinned buffers go back to the ring."""
        slot = getattr(packed, 'slot', None)
        if slot is not None:
            self._oring.release(slot)
            packed.slot = None

    # ------------------------------------------------------------------
    # Certification of the rounding and replay in the reference's order
    # (certify.py): the device sums exactly, the reference sums binary64
    # numbers chunk by chunk; where the two might round differently the cell
    # is summed again the reference's way.
    def begin_file(self):
        """A new alignment file starts: the reference's mapper chunks count
        queries per file (align.plain_mapper, align.py:84-115)."""
        self._spec = False
        if self._replay is not None:
            self._replay_close()
            self._replay['pos'] = 0
        else:
            self._n_files += 1

    # ------------------------------------------------------------------
    def _n_batches(self):
        return (len(self.jobs) + nat.MAX_JOBS - 1) // nat.MAX_JOBS

    def _classify_staged(self, data, want):
        """All ranks over the staged chunk.  One launch carries at most
        WK_MAX_JOBS jobs (3 key bits); more ranks (`--rank a,b,...` has no
        limit in the reference, workflow.py:333-335) run in batches over the
        same staged chunk, the counts of a batch folded to the host before the
        next one reuses the job numbers."""
        if len(self.jobs) <= nat.MAX_JOBS:
            return self.ctx.classify_staged(self.jobs, want_assign=want)
        parts = []
        self.collect(data, keep_groups=True)
        for lo in range(0, len(self.jobs), nat.MAX_JOBS):
            self._job_base = lo
            try:
                parts.append(self.ctx.classify_staged(
                    self.jobs[lo:lo + nat.MAX_JOBS], want_assign=want))
                if self.sizes:
                    self._collect_log()
                self.collect(data, keep_groups=True)
            finally:
                self._job_base = 0
        return np.concatenate(parts) if want else None

    def _rank_table(self, slot):
        if slot not in self._anc:
            self._anc[slot] = self.ctx.get_rank_table(slot)
        return self._anc[slot]
