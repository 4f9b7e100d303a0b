"""Packed records accumulated per sample (wk_words_*): the weighted histogram
for plain ranks (csrc/wk_weigh.hpp) and the per-read stream for `--rank free`
/ `--uniq` / `--above` / `--major` (csrc/wk_free.hpp); which job sets they take."""
import os

import numpy as np

from .. import _native as nat
from ..hostio import MAX_GROUPS


class WordsRoute:
    """(mixin of classify.Engine)"""

    def words_eligible(self, identity=True):
        """Can chunks go to the device as packed words, accumulated per
        sample (``wk_words_*``)?  The plain assigners only — what
        ``wk_words_begin`` checks once more against the subject table.
        ``identity``: the words come from the host tokenizer, whose ids must
        be the subject indices (the device text route translates)."""
        if self.sizes or self._replay is not None or \
                len(self.jobs) > nat.MAX_JOBS or \
                (identity and not self._tok_identity) or \
                os.environ.get('WOLTKA_NO_WORDS'):
            return False
        # jobs that look at whole reads — `--rank free`, a rank under --uniq /
        # --above / --major above one half — all go to the per-read stream
        # (csrc/wk_free.hpp), alone or several of them
        def whole_reads(job):
            if job.flags & nat.F_SIZED:
                return False
            if job.mode == nat.MODE_FREE:
                return True
            return job.mode == nat.MODE_RANK and (
                job.major > 0.5 or (job.major <= 0 and bool(
                    job.flags & (nat.F_UNIQ | nat.F_ABOVE))))
        if all(map(whole_reads, self.jobs)):
            return self.use_tree
        for job in self.jobs:
            if job.flags & (nat.F_UNIQ | nat.F_SIZED):
                return False
            if job.mode == nat.MODE_RANK and (job.flags & nat.F_ABOVE or
                                              job.major > 0):
                return False
            if job.mode not in (nat.MODE_NONE, nat.MODE_RANK):
                return False
        return True

    def device_maps_eligible(self):
        """Can the read maps be formatted on the device (wk_readmap.hpp)?  The
        plain assigners, i.e. the job sets the weighted histogram takes."""
        if self.sizes or self._replay is not None or \
                len(self.jobs) > nat.MAX_JOBS or not self._tok_identity or \
                os.environ.get('WOLTKA_NO_WORDS') or \
                os.environ.get('WOLTKA_NO_DMAPS'):
            return False
        for job in self.jobs:
            if job.flags & (nat.F_UNIQ | nat.F_SIZED | nat.F_ABOVE) or \
                    job.major > 0 or \
                    job.mode not in (nat.MODE_NONE, nat.MODE_RANK):
                return False
        return True

    def _run_words(self, data, packed, sample):
        """One chunk of packed records (``('words', array, n_reads, slot)``
        from `native_chunks`): appended to the sample's records on the device,
        which are classified by one launch when the sample ends
        (``wk_words_flush`` — any fetch of the counts flushes)."""
        _, words, n, slot = packed
        ring = self._ring
        if (sample, None) not in self.group_ids:
            # a new sample: room for its group id and for the keys it can add
            # (one per job and taxon, at most one per subject) — checked once
            # per sample, not per chunk: looking at the table waits for the
            # device
            if len(self.groups) + 1 >= MAX_GROUPS // 2:
                self.collect(data)
            self._ensure_table(data, max(len(self.subjects), 1 << 16) + 1, 1)
        group = self._group_array(n, sample, None)
        for rank in self.ranks:
            data[rank].setdefault(sample, {})
        self._n_reads += n
        known = len(self.subj_feature)
        if len(self.subjects) > known:
            self.subj_feature.extend(self.index.intern_many(
                self.subjects.names[known:]))
            self.ctx.set_subjects(self.subj_feature)
            # (a key per job and subject at most; the table is never left to
            # fill up)
            if 4 * len(self.subjects) * len(self.jobs) > self.slots_reserved \
                    and not self._table_fixed:
                self.collect(data, keep_groups=True)
                self._reserve(8 * len(self.subjects) * len(self.jobs))
        if self.ctx.words_begin(self.jobs, group):
            self.ctx.words_append(words, n, slot)
            # the buffer of the chunk before this one has been copied by now
            if self._ring_prev is not None:
                self.ctx.words_wait(self._ring_prev)
                ring.release(self._ring_prev)
            self._ring_prev = slot
            return n
        # the general route (a subject without an ancestor at some rank):
        # subject indices and read offsets out of the words
        w = np.array(words)                 # (off the pinned buffer)
        ring.release(slot)
        subj = (w & np.uint32((1 << nat.Context.WORD_SUBJ_BITS) - 1)
                ).astype(np.int32)
        starts = np.flatnonzero((w >> np.uint32(nat.Context.WORD_POS_SHIFT)) &
                                np.uint32(15) == 0)
        qoff = np.concatenate((starts, [w.size])).astype(np.int32)
        self.ctx.chunk_stage(subj, qoff, group=group, subj_is_set=True,
                             indexed=True)
        self._classify_staged(data, False)
        return n

    def _words_done(self):
        """The last staging buffer in flight goes back to the ring."""
        if self._ring_prev is not None:
            self.ctx.words_wait(self._ring_prev)
            self._ring.release(self._ring_prev)
            self._ring_prev = None
