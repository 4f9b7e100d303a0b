"""`woltka classify` workflow on the MI355X path.

Drop-in counterpart of the reference's ``woltka/workflow.py``: ``workflow()``
(:44-159) and ``classify()`` (:162-353) keep their signatures, their console
output and the shape of the returned ``data`` dict
(``{rank: {sample: {feature or (stratum, feature): number}}}``).  What changes is
*where* the per-read work happens: instead of ``assign_readmap`` looping over
Python tuples (:941-1058), every chunk of alignments is packed into integer
arrays (``align.pack_queries`` / ``ordinal.pack_hits``) and handed to the HIP
kernels through ``classify.Engine``; counts come back as exact integers and are
folded into the same dict.

There is no CPU fallback: ``classify()`` needs ``libwoltka_hip.so`` and a GPU.
"""
import contextlib
import io
import os
from functools import partial
from itertools import chain
from os import makedirs
from os.path import basename, isdir, isfile, join

import click
import numpy as np

from .align import NATIVE_FORMATS, infer_align_format, plain_mapper
from .classify import Engine, exact_to_numbers
from .file import (FilesAhead, id2file_from_dir, id2file_from_map, openzip, path2stem,
                   read_ids, read_map_1st, read_map_uniq, readzip, readzip_bytes,
                   stem2rank, write_readmap)
from .ordinal import load_gene_coords, load_gene_coords_file  # noqa: F401
from .ranges import Coverage, range_mapper, write_coverage
from .shard import (FilePart, classify_sharded, file_key,
                    file_path)
from .table import allkeys, prep_table, write_table
from .tree import (fill_root, read_columns, read_lineage, read_names,
                   read_newick, read_nodes)

DEVICE_CHUNK = 2 ** 20      # queries per device chunk unless --chunk is given
# bytes of alignment text per native tokenizer call (= per staged chunk): 256 MiB
# halves the per-block host work of 128 MiB (config 3 end to end: 217 -> 259 M
# records/s while streaming); the environment variable is a measurement knob
NATIVE_BLOCK = int(os.environ.get('WOLTKA_NATIVE_BLOCK', 1 << 28))


class OrdinalMapper:
    """Coord-match mapper: gene table + overlap threshold.  ``classify()``
    recognises it and runs match + assignment on the device without a round
    trip; the gene table is uploaded once per job (``Engine.set_genes``)."""

    def __init__(self, table, th=0.8, prefix=None):
        self.table = table
        self.th = th
        self.prefix = table.isdup if prefix is None else prefix


def _update(dic, other):
    """util.update_dict (woltka/util.py:46-75): merge, conflicting values are
    an error."""
    for key, value in other.items():
        if key in dic:
            assert dic[key] == value, f'Conflicting values found for "{key}".'
        else:
            dic[key] = value


def workflow(
        input_fp: str, output_fp: str, input_fmt: str = None,
        input_ext: str = None, samples: str = None, demux: bool = None,
        exclude: set = None, trimsub: str = None, nodes_fps: list = [],
        newick_fps: list = [], lineage_fps: list = [], columns_fps: list = [],
        map_fps: list = [], map_rank: bool = False, names_fps: list = [],
        ranks: str = None, uniq: bool = False, major: bool = None,
        above: bool = False, subok: bool = False, coords_fp: str = None,
        overlap: int = 80, strata_dir: str = None, sizes: str = None,
        frac: bool = False, scale: str = None, digits: int = None,
        output_fmt: str = None, unassigned: bool = False,
        name_as_id: bool = False, add_rank: bool = False,
        add_lineage: bool = False, outmap_dir: str = None,
        outmap_zip: str = 'gz', outcov_dir: str = None,
        outcov_fmt: str = None, chunk: int = None, cache: int = 1024,
        no_exe: bool = False, device: int = 0, gpus: int = 1,
        comm: object = None) -> dict:
    """Main classification workflow (command-line arguments in, profile out);
    same steps in the same order as the reference (workflow.py:109-159)."""
    # (the run builds millions of small containers — table rows, profile
    # cells — and no reference cycles: the cyclic collector's passes over them
    # cost more than the classification; paused for the call)
    # (and the helper threads — readers, formatters — hold the interpreter for
    # moments only, but this thread asks for it back after every native call:
    # with the default switch interval of 5 ms a hand-over cost ~0.2 ms, a
    # dozen times per block of text)
    import gc
    import sys
    gc_was_on = gc.isenabled()
    switch = sys.getswitchinterval()
    gc.disable()
    sys.setswitchinterval(min(switch, 1e-4))
    try:
        return _workflow(**{k: v for k, v in locals().items()
                            if k not in ('gc', 'gc_was_on', 'sys', 'switch')})
    finally:
        sys.setswitchinterval(switch)
        if gc_was_on:
            gc.enable()


def _workflow(input_fp, output_fp, input_fmt, input_ext, samples, demux,
              exclude, trimsub, nodes_fps, newick_fps, lineage_fps,
              columns_fps, map_fps, map_rank, names_fps, ranks, uniq, major,
              above, subok, coords_fp, overlap, strata_dir, sizes, frac, scale,
              digits, output_fmt, unassigned, name_as_id, add_rank,
              add_lineage, outmap_dir, outmap_zip, outcov_dir, outcov_fmt,
              chunk, cache, no_exe, device, gpus=1, comm=None):
    # `--gpus N`: this process becomes rank 0 of N and starts the others
    # (shard.LocalWorld: multiprocessing, no PyTorch).  Ranks somebody else
    # started (mpirun, torch.distributed.run ...) hand in their own `comm`:
    # any object with rank / local / world / kind and gather(obj) -- the
    # package itself never looks at a launcher's environment
    procs = []
    if comm is None and gpus and gpus > 1:
        from .shard import start_local_world
        kw = {k: v for k, v in locals().items()
              if k not in ('comm', 'procs', 'gpus', 'start_local_world')}
        comm, procs = start_local_world(gpus, _rank_entry, kw)
    failed = True
    try:
        out = _workflow_ranked(
            input_fp, output_fp, input_fmt, input_ext, samples, demux,
            exclude, trimsub, nodes_fps, newick_fps, lineage_fps,
            columns_fps, map_fps, map_rank, names_fps, ranks, uniq, major,
            above, subok, coords_fp, overlap, strata_dir, sizes, frac, scale,
            digits, output_fmt, unassigned, name_as_id, add_rank,
            add_lineage, outmap_dir, outmap_zip, outcov_dir, outcov_fmt,
            chunk, cache, no_exe, device, comm)
        failed = False
        return out
    finally:
        if procs:
            # (a failed run must not wait on ranks blocked in send())
            from .shard import stop_local_world
            stop_local_world(comm, procs, failed=failed)


def _rank_entry(comm=None, **kw):
    """A rank `woltka classify --gpus N` started (shard.start_local_world)."""
    return _workflow(gpus=1, comm=comm, **kw)


def _workflow_ranked(input_fp, output_fp, input_fmt, input_ext, samples, demux,
                     exclude, trimsub, nodes_fps, newick_fps, lineage_fps,
                     columns_fps, map_fps, map_rank, names_fps, ranks, uniq,
                     major, above, subok, coords_fp, overlap, strata_dir,
                     sizes, frac, scale, digits, output_fmt, unassigned,
                     name_as_id, add_rank, add_lineage, outmap_dir,
                     outmap_zip, outcov_dir, outcov_fmt, chunk, cache, no_exe,
                     device, comm):
    zippers = None if no_exe else {}
    samples, files, demux = parse_samples(input_fp, input_ext, samples, demux)
    exclude = parse_exclude(exclude)
    stratmap = parse_strata(strata_dir, samples)
    # (the device context comes up while the inputs below are read)
    from . import classify as _classify
    restore = None
    if comm is None:
        # (with the pinned buffers of the device text route, when the input
        # can take it: plain / gzip text without an exclusion set)
        ring = None
        if not exclude and not os.environ.get('WOLTKA_NO_DTOK') and \
                not os.environ.get('WOLTKA_NO_PIN_AHEAD'):
            # (only for inputs that will use all of them: a run of a few
            # hundred MB is over before eight buffers are pinned)
            try:
                total = sum(os.path.getsize(file_path(x)) for x in files
                            if file_path(x) != '-')
            except OSError:
                total = 0
            if total >= (2 << 30):
                from .routes.device_text import DeviceTextRoute as _R
                ring = (8, _R.DTOK_BLOCK + _R.DTOK_HEADROOM)
        # (--stratify: two pinned buffers for the text of the samples' read
        # maps, sized for the largest)
        strata_bytes = 0
        if stratmap and not os.environ.get('WOLTKA_NO_DTOK') and \
                not os.environ.get('WOLTKA_NO_PIN_AHEAD'):
            from .routes.device_text import DeviceTextRoute as _R
            try:
                for fp in stratmap.values():
                    need = os.path.getsize(fp)
                    if fp.endswith('.gz'):
                        need = _R._inflated_size(fp) or need * 8
                    strata_bytes = max(strata_bytes, need)
            except OSError:
                strata_bytes = 0
            if strata_bytes:
                strata_bytes = int(strata_bytes * 1.25) + (1 << 20)
            if strata_bytes > (8 << 30):
                strata_bytes = 0
        _classify.open_context_ahead(device, ring, strata_bytes)
    elif comm.kind == 'local':
        # the ranks share this node's CPUs (classify.tokenizer_threads) and
        # each keeps to the NUMA node of its GPU
        from . import _native as nat
        from .shard import pin_near_gpu
        restore = (os.environ.get('LOCAL_WORLD_SIZE'),
                   os.sched_getaffinity(0))
        os.environ['LOCAL_WORLD_SIZE'] = str(comm.world)
        pin_near_gpu(comm.local % max(nat.device_count(), 1))
    try:
        return _workflow_with_context(
            samples, files, demux, exclude, stratmap, zippers, output_fp,
            input_fmt, trimsub, nodes_fps, newick_fps, lineage_fps,
            columns_fps, map_fps, map_rank, names_fps, ranks, uniq, major,
            above, subok, coords_fp, overlap, sizes, frac, scale, digits,
            output_fmt, unassigned, name_as_id, add_rank, add_lineage,
            outmap_dir, outmap_zip, outcov_dir, outcov_fmt, chunk, cache,
            device, comm)
    finally:
        _classify.drop_context_ahead()
        if restore is not None:
            if restore[0] is None:
                os.environ.pop('LOCAL_WORLD_SIZE', None)
            else:
                os.environ['LOCAL_WORLD_SIZE'] = restore[0]
            try:
                os.sched_setaffinity(0, restore[1])
            except OSError:
                pass


def _workflow_with_context(
        samples, files, demux, exclude, stratmap, zippers, output_fp,
        input_fmt, trimsub, nodes_fps, newick_fps, lineage_fps, columns_fps,
        map_fps, map_rank, names_fps, ranks, uniq, major, above, subok,
        coords_fp, overlap, sizes, frac, scale, digits, output_fmt,
        unassigned, name_as_id, add_rank, add_lineage, outmap_dir, outmap_zip,
        outcov_dir, outcov_fmt, chunk, cache, device, comm=None):
    # (while the hierarchy and the gene coordinates are read: the subjects of
    # the first alignment file's first bytes into the tokenizer's dictionary,
    # so that the device text route starts with full blocks and few unknowns)
    if files and not exclude and (comm is None or comm.world == 1) and \
            not os.environ.get('WOLTKA_NO_WARM'):
        from .shard import file_key, file_path, FilePart
        from .file import ZIP_BY_EXT
        from os.path import isfile, splitext
        first = sorted(files, key=file_key)[0]
        fp0 = file_path(first)
        if not isinstance(first, FilePart) and fp0 != '-' and isfile(fp0) \
                and ZIP_BY_EXT.get(splitext(fp0)[1]) is None and \
                input_fmt in (None, 'sam', 'b6o', 'paf', 'map'):
            from . import classify as _classify
            _classify.warm_tokenizer_ahead(fp0, input_fmt)
            # ... and the file's blocks on their way to the device (the
            # context is being created on its thread: open_context_ahead)
            # (not under --stratify: reading and joining the first sample's
            # read map wants the CPUs the reader would take -- config 5's
            # second call lost 0.25 s of 2.15 to it)
            if comm is None and not stratmap and \
                    not os.environ.get('WOLTKA_NO_DTOK'):
                from .routes.device_text import start_text_ahead
                start_text_ahead(fp0, input_fmt, device,
                                 extra=bool(coords_fp),
                                 ordered=bool(outmap_dir))
    start_coords_ahead(coords_fp, zippers)
    try:
        tree, rankdic, namedic, root = build_hierarchy(
            names_fps, nodes_fps, newick_fps, lineage_fps, columns_fps,
            map_fps, map_rank, zippers)
    except BaseException:
        _coords_ahead.clear()
        raise
    mapper, chunk = build_mapper(coords_fp, outcov_dir, overlap, chunk,
                                 zippers)
    sizes = parse_sizes(sizes, mapper, zippers)
    ranks, rank2dir = prepare_ranks(ranks, outmap_dir, tree, rankdic)

    def run(share, dev, exact=False):
        return classify(
            mapper, share, samples, input_fmt, demux, trimsub, tree, rankdic,
            namedic if name_as_id else None, root, ranks, rank2dir,
            outmap_zip, uniq, major, above, subok, sizes, unassigned, stratmap,
            exclude, chunk, cache, zippers, outcov_dir, outcov_fmt, device=dev,
            exact=exact,
            rounding=(digits, scale_factor(scale) if scale else None, frac))

    # one process per GPU (`--gpus N`, or ranks that came with their `comm`):
    # alignment files (samples) shard across processes, profiles merge on the
    # host
    if comm is not None and comm.world > 1:
        from . import _native as nat
        dev = comm.local % max(nat.device_count(), 1)   # narrowed visibility: 0
        # one large plain file may be cut into byte ranges, unless per-sample
        # side files (read maps, coverage) are written; cells of a sample
        # that several processes saw are added as exact rationals
        data = classify_sharded(lambda share: run(share, dev, exact=True),
                                files, comm.rank, comm.world,
                                gather=comm.gather,
                                split=not (outmap_dir or outcov_dir))
        if data is None:        # (gathered on rank 0 only)
            return None
        exact_to_numbers(data)
        for r in ranks:
            data.setdefault(r, {})
        if comm.rank != 0:
            return data
    else:
        data = run(files, device)
    frac_profiles(data, frac)
    scale_profiles(data, scale)
    round_profiles(data, digits)
    write_profiles(data, output_fp, output_fmt, samples, tree, rankdic,
                   namedic, name_as_id, add_rank, add_lineage)
    click.echo('Task completed.')
    return data


def classify(
        mapper: object, files: list or dict, samples: list = None,
        fmt: str = None, demux: bool = None, trimsub: str = None,
        tree: dict = None, rankdic: dict = None, namedic: dict = None,
        root: str = None, ranks: str = None, rank2dir: dict = None,
        outzip: str = None, uniq: bool = False, major: int = None,
        above: bool = False, subok: bool = False, sizes: dict = None,
        unasgd: bool = False, stratmap: dict = None, exclude: set = None,
        chunk: int = None, cache: int = 1024, zippers: dict = None,
        outcov_dir: str = None, outcov_fmt: str = None, device: int = 0,
        exact: bool = False, rounding: tuple = None) -> dict:
    """Core of the classification workflow (workflow.py:162-353) on the GPU.

    ``mapper`` is ``align.plain_mapper`` (or any generator with the reference's
    mapper protocol, workflow.py:304) or an ``OrdinalMapper``.  ``cache`` (the
    reference's per-rank LRU size) is accepted and ignored: nothing is
    memoised, every read is evaluated by the kernels.  Counts are accumulated
    as exact integers on the device, so the result does not depend on ``chunk``
    — except where the reference's own float summation decides a rounding:
    ``rounding`` = (digits, scale factor, frac) tells which rounding follows
    (default: to integers), and cells that are not certain to round like the
    reference's are summed again in its order, ``chunk`` (default 1024) queries
    at a time (certify.py).
    """
    data = {x: {} for x in ranks}
    cover = Coverage() if outcov_dir else None
    outzip = outzip if outzip != 'none' else None
    engine = Engine(tree, rankdic, root, ranks, uniq=uniq, major=major,
                    above=above, subok=subok, unasgd=unasgd, device=device,
                    sizes=sizes)
    ordinal = isinstance(mapper, OrdinalMapper)
    if ordinal:
        engine.set_genes(mapper.table, mapper.prefix, trimsub,
                         read_maps=rank2dir is not None)
    n = chunk or DEVICE_CHUNK
    csample, strata = False, None
    try:
        # SAM, BLAST tabular, PAF and simple map input go through the native
        # multi-threaded tokenizer (every flavour: plain / "extra", with or
        # without an exclusion set — the reference's SAM parser for
        # "extra + exclude" yields its pool once more at the end of a file
        # whose last query was dropped, align.py:542-547: the tokenizer tracks
        # that pool, align._tail_block); anything else uses the Python
        # parsers.  Query names are materialised only when something needs
        # them.
        if cover is not None and ordinal:
            raise ValueError('Subject coverage (--outcov) needs subject-level '
                             'alignments; it cannot be combined with --coords.')
        native_ok = (mapper is plain_mapper or mapper is range_mapper or
                     ordinal)
        # stratification without demultiplexing is joined natively (read id ->
        # stratum inside the tokenizer)
        native_strata = bool(stratmap) and not demux
        # demultiplexing alone (no strata, no read maps) is done natively too
        native_demux = bool(demux) and not stratmap and rank2dir is None \
            and cover is None
        allow = set(samples) if (demux and samples) else None
        engine._exclude = exclude
        labels = None

        def one_pass(rank2dir, cover):
            """All files through the device once (the main pass; the replay of
            uncertified cells runs it a second time)."""
            nonlocal labels
            labels = None
            order = sorted(files, key=file_key)
            # compressed inputs: the next files are inflated while this one is
            # tokenised and classified
            ahead = FilesAhead([file_path(x) for x in order], zippers)
            try:
                files_loop(order, ahead, rank2dir, cover)
            finally:
                ahead.close()

        def files_loop(order, ahead, rank2dir, cover):
            nonlocal csample, strata, labels
            for ifile, fp in enumerate(order):
                # (a FilePart is one of several byte ranges of a large file that
                # other processes share, shard.partition_files)
                path = file_path(fp)
                part = (fp.part, fp.parts) if isinstance(fp, FilePart) else None
                engine.begin_file()
                if path == '-':
                    stream = click.get_binary_stream('stdin')
                    click.echo('Parsing alignment from stdin ', nl=False)
                else:
                    stream = ahead.open(ifile)
                    click.echo(f'Parsing alignment file {basename(path)} ',
                               nl=False)
                with stream:
                    nqry, nstep = 0, -1
                    fmt_, head = fmt, b''
                    if not fmt_:
                        head = stream.readline()
                        fmt_ = infer_align_format(iter(
                            [head.decode()] if head else []))[0]
                    native = native_ok and fmt_ in NATIVE_FORMATS and not (
                        fmt_ == 'map' and (ordinal or cover is not None))
                    if part is not None and not native:
                        # byte ranges are a feature of the native tokenizer: the
                        # first part takes the whole file, the others nothing
                        if part[0]:
                            click.echo(' Done.')
                            continue
                        part = None
                    want_names = bool((demux and not native_demux) or
                                      rank2dir is not None or
                                      (stratmap and not (native and native_strata)))
                    if native:
                        if native_strata:
                            sample = files[fp] if files else None
                            if sample != csample or labels is None:
                                # (the map of the sample after this one is
                                # read meanwhile)
                                later = [files[x] for x in files]
                                later = [x for x in later[later.index(sample):]
                                         if x != sample and x in stratmap]
                                # coord-match on text that the device tokenises
                                # (SAM, BLAST tabular, PAF): the join runs
                                # there too
                                from .file import ZIP_BY_EXT, GunzipStream
                                from os.path import splitext
                                dstrata = bool(
                                    ordinal and fmt_ in ('sam', 'b6o', 'paf')
                                    and not exclude
                                    and part is None and cover is None and
                                    rank2dir is None and path != '-' and
                                    (ZIP_BY_EXT.get(splitext(path)[1]) is None
                                     or isinstance(stream, GunzipStream))
                                    and not os.environ.get('WOLTKA_NO_DTOK'))
                                labels = engine.load_strata(
                                    stratmap[sample], zippers,
                                    then=stratmap[later[0]] if later else None,
                                    device=dstrata)
                                csample = sample
                        # read ids as Python strings only when the host logic
                        # needs them (demultiplexing, Python-side strata join);
                        # read maps alone are formatted natively from descriptors
                        # (coord-match read maps: the reference's listing
                        # order is worked out on the host, per mapper chunk)
                        want_strings = bool((demux and not native_demux) or (
                            stratmap and not native_strata) or (
                            ordinal and rank2dir is not None))
                        # plain assigners, one sample per file, nothing per
                        # read: the records cross as packed words and the
                        # sample is classified by one launch at its end
                        plain = not (ordinal or cover is not None or
                                     want_names or native_strata or demux or
                                     rank2dir is not None)
                        words = plain and not trimsub and \
                            engine.words_eligible()
                        # (`--trim-sub`: the device text route translates the
                        # names it meets into subjects; the host tokenizer's
                        # words cannot)
                        # (`--exclude`: the names of the set get no subject
                        # index, so the tokenizer's ids are none either)
                        words_dev = plain and bool(trimsub or exclude) and \
                            engine.words_eligible(identity=False)
                        # read maps of plain assigners, one sample per file:
                        # the lines are formatted on the device next to the
                        # tokenised text (csrc/wk_readmap.hpp)
                        dmaps = None
                        if rank2dir is not None and not (
                                ordinal or cover is not None or demux or
                                stratmap or trimsub or want_strings) and \
                                engine.device_maps_eligible():
                            dmaps = (rank2dir, outzip, namedic)
                        chunks = engine.native_chunks(
                            stream, head, exclude, NATIVE_BLOCK, ordinal,
                            want_names, trimsub, want_groups=native_strata,
                            want_strings=want_strings, want_samples=native_demux,
                            cover=cover, fmt=fmt_, part=part, words=words,
                            words_dev=words_dev, dmaps=dmaps,
                            keep_empty=bool(ordinal and rank2dir is not None))
                        if ordinal and rank2dir is not None:
                            chunks = engine.regroup_hits(chunks, n)
                    else:
                        text = io.TextIOWrapper(
                        io.BufferedReader(stream) if isinstance(
                            stream, io.RawIOBase) and not isinstance(
                            stream, io.BufferedIOBase) else stream,
                        encoding='utf-8')
                        fh = chain([head.decode()], text) if head else text
                        if ordinal:
                            chunks = engine.ordinal_chunks(fh, fmt_, exclude, n,
                                                           mapper.th)
                        else:
                            chunks = mapper(fh, fmt=fmt_, excl=exclude, n=n)
                    for chunk_ in chunks:
                        packed = strata_ids = names = sample_ids = None
                        if native:
                            qryque, packed, strata_ids, names, sample_ids, \
                                ranges = chunk_
                            subque = None
                            engine._th = mapper.th if ordinal else None
                        elif ordinal:
                            qryque = chunk_
                            subque = None
                        else:
                            qryque, subque = chunk_
                        # sample of every read (demultiplexing / whitelist)
                        if demux and sample_ids is not None:
                            sample_of, reads = None, None
                        elif demux:
                            sample_of, reads = demux_labels(qryque, samples)
                        else:
                            sample_of = files[fp] if files else None
                            reads = qryque
                        # (optional) aligned ranges per (sample, subject)
                        # (parse_ranges, workflow.py:312)
                        if cover is not None and native:
                            if demux:
                                per_read = np.fromiter(
                                    (-1 if x is False else cover.sample(x)
                                     for x in sample_of), np.int64, len(sample_of))
                                who = np.repeat(per_read, np.diff(packed[-1]))
                            else:
                                who = cover.sample(sample_of)
                            cover.add(who, *ranges)
                        elif cover is not None:
                            cover.add_queries(sample_of, subque)
                        # stratum of every read; the strata map of a sample is read
                        # when the sample first shows up (workflow.py:327-330)
                        strata_of = None
                        if stratmap and strata_ids is None and not (
                                native and native_strata):
                            strata_of, csample, strata = strata_labels(
                                sample_of, reads, stratmap, zippers, csample,
                                strata)
                        nq = engine.run_chunk(
                            data, reads, subque, sample_of, strata_of,
                            None if native else trimsub,
                            rank2dir, outzip, namedic, ordinal, packed=packed,
                            strata_ids=strata_ids, strata_labels=labels,
                            names=names, sample_ids=sample_ids, allow=allow,
                            packed_is_set=not trimsub and cover is None)
                        nqry += nq
                        istep = nqry // 1000000 - nstep
                        if istep:
                            click.echo('.' * istep, nl=False)
                            nstep += istep
                    # (queries the device counted and `run_chunk` has not
                    # reported: the dots they are owed come now — the text is
                    # the same, a dot per million and one at the first chunk)
                    nqry += engine.take_deferred()
                    if nstep >= 0:      # (at least one chunk was classified)
                        click.echo('.' * (nqry // 1000000 - nstep), nl=False)
                click.echo(' Done.')
                click.echo(f'  Number of sequences classified: {nqry}.')

        one_pass(rank2dir, cover)
        engine.finish(data, exact)
        # Cells whose exact value lies so close to a rounding boundary that
        # the reference's float summation might land on the other side are
        # summed once more in the reference's own order (certify.py).
        # Under torch.distributed (`exact`: the profiles of the processes
        # are added as exact rationals) a process certifies the samples it
        # holds whole — only it sees their records; a sample cut into byte
        # ranges over several processes (FilePart) or demultiplexed out of
        # files of several processes cannot be replayed by one of them and
        # keeps its exact value.
        whole = None
        if exact:
            whole = set()
            if isinstance(files, dict) and not demux:
                cut = {s for fp, s in files.items() if isinstance(fp, FilePart)}
                whole = {s for fp, s in files.items()
                         if not isinstance(fp, FilePart)} - cut
        if (not exact or whole) and not sizes and not ordinal and \
                cover is None and mapper is plain_mapper and \
                '-' not in map(file_path, files):
            digits, factor, frac = rounding or (None, None, False)
            todo = {} if frac else engine.uncertified(
                digits, factor, chunk or 1024)
            if whole is not None:
                todo = {r: {s: k for s, k in per.items() if s in whole}
                        for r, per in todo.items()}
                todo = {r: per for r, per in todo.items() if per}
            if todo:
                engine.replay_begin(todo, chunk or 1024)
                csample, strata = False, None
                with contextlib.redirect_stdout(io.StringIO()):
                    one_pass(None, None)
                for (rank, sample, key), value in engine.replay_end().items():
                    data[rank][sample][key] = value
    finally:
        engine.close_later()
    if cover is not None:
        click.echo('Calculating per sample coverage...', nl=False)
        write_coverage(cover.merged(), outcov_dir, outcov_fmt)
        click.echo(' Done.')
    click.echo('Classification completed.')
    return data


def assign_readmap(qryque, subque, data, rank, sample, assigners, cache=1024,
                   rank2dir=None, outzip=None, tree=None, rankdic=None,
                   namedic=None, root=None, uniq=False, major=None,
                   above=False, subok=False, sizes=None, unasgd=False,
                   strata=None):
    """One chunk of queries at one rank: classify, write the read map if asked
    for, and add the counts into ``data[rank][sample]`` — the reference's
    per-chunk entry point (workflow.py:941-1058) on the GPU.

    ``classify()`` above does not go through it (it keeps all ranks of a chunk
    in one launch and the counts on the device until the end); it is here for
    callers that drive the reference chunk by chunk.  ``assigners`` caches one
    device engine per rank, like the reference caches one assigner per rank;
    ``cache`` (an LRU size) has no counterpart.  ``major`` is the fraction, as
    in the reference.  Counts are exact: a cell that the reference holds as a
    float sum of 1/k terms is an ``int`` when integral, else the correctly
    rounded quotient."""
    plain = rank is None or rank == 'none' or tree is None
    key = 'none' if plain else rank
    engine = assigners.get(key)
    if engine is None:
        engine = assigners[key] = Engine(
            None if plain else tree, rankdic, root, [key], uniq=uniq,
            above=above, subok=subok, unasgd=unasgd, sizes=sizes,
            major_frac=major)
    qryque, subque = list(qryque), list(subque)
    strata_of = [strata.get(q) for q in qryque] if strata else None
    part = {key: {}}
    engine.run_chunk(part, qryque, subque, sample, strata_of, None,
                     None if rank2dir is None else {key: rank2dir[rank]},
                     outzip, namedic, False)
    engine.finish(part)
    cells = data[rank].setdefault(sample, {})
    for feature, value in part[key].get(sample, {}).items():
        cells[feature] = cells[feature] + value if feature in cells else value


def demux_labels(qryque, samples=None, sep='_'):
    """workflow.demultiplex (workflow.py:844-909) as per-read labels: returns
    (sample of each read or ``False`` when dropped, read ids).  A query splits
    at the first separator into (sample, read); without separator (or with an
    empty right part) the sample is ''/None-like and the read keeps the whole
    left part; the whitelist is consulted only when the sample changes."""
    allow = set(samples) if samples else None
    labels, reads = [], []
    cur = False
    for query in qryque:
        left, _, right = query.partition(sep)
        sample, read = (right and left), (right or left)
        if sample == cur:
            labels.append(cur)
        elif allow is None or sample in allow:
            cur = sample
            labels.append(sample)
        else:
            labels.append(False)
        reads.append(read)
    return labels, reads


def demultiplex(qryque, subque, samples=None, sep='_'):
    """{sample: (read ids, subject(s) queue)} of a multiplexed chunk, samples
    in order of first appearance (workflow.py:844-909); the per-read form the
    device path uses is ``demux_labels``."""
    labels, reads = demux_labels(qryque, samples, sep)
    res = {}
    for label, read, subjects in zip(labels, reads, subque):
        if label is not False:
            ids, subs = res.setdefault(label, ([], []))
            ids.append(read)
            subs.append(subjects)
    return res


def strip_suffix(subque, sep):
    """Subject ids cut at their last ``sep``, every query's subjects as a set
    again (workflow.py:818-841); the packers do this while interning
    (``pack_queries(trim=)``, the tokenizer's ``trimsub``)."""
    return ({x.rsplit(sep, 1)[0] for x in subjects} for subjects in subque)


def strata_labels(sample_of, reads, stratmap, zippers, csample, strata):
    """Stratum of every read (``None`` = not in the strata map, skipped by the
    counters, classify.py:239).  Strata files are (re)read when the current
    sample changes, in read order like the reference does chunk by chunk."""
    out = []
    if isinstance(sample_of, list):
        for s, read in zip(sample_of, reads):
            if s is False:
                out.append(None)
                continue
            if s != csample:
                strata = read_strata(stratmap[s], zippers)
                csample = s
            out.append(strata.get(read))
    else:
        if sample_of != csample:
            strata = read_strata(stratmap[sample_of], zippers)
            csample = sample_of
        get = strata.get
        out = [get(r) for r in reads]
    return out, csample, strata


INCONSISTENT = 'Provided sample IDs and actual files are inconsistent.'


def _id_list(arg):
    """Ids given as a comma-separated string or as the first column of a
    (possibly compressed) list file."""
    if not isfile(arg):
        return arg.split(',')
    with openzip(arg) as fh:
        return read_ids(fh)


def parse_samples(fp, ext=None, samples=None, demux=None):
    """Sample ids, alignment files and demultiplexing switch
    (workflow.py:356-480).  Four kinds of input — stdin, a directory, a
    sample-to-file table, one alignment file — each decide what the default of
    ``demux`` means and whether ``files`` is a list (to demultiplex) or a
    {path: sample} dict."""
    wanted = None
    if samples:
        wanted = _id_list(samples)
        click.echo(f'Number of samples to include: {len(wanted)}.')

    def single(path, sample, note):
        # one stream: demultiplex unless told not to (--no-demux)
        on = demux is not False
        if not on and wanted and wanted != [sample] and path != '-':
            raise ValueError(INCONSISTENT)
        click.echo(note)
        if on:
            return wanted, [path], True
        return [sample], {path: sample}, False

    if fp == '-':
        res = single(fp, '', 'Input alignment is from stdin.')
    elif isdir(fp):
        on = bool(demux)
        found = id2file_from_dir(fp, ext, not on and wanted)
        if not found:
            raise ValueError('No valid file found in directory.')
        if on:
            files = sorted(join(fp, name) for name in found.values())
        else:
            if not wanted:
                wanted = sorted(found)
            elif len(found) < len(wanted):
                raise ValueError(INCONSISTENT)
            files = {join(fp, found[x]): x for x in wanted}
        click.echo(f'Input directory: {fp}.')
        click.echo(f'Number of alignment files to read: {len(files)}.')
        res = wanted, files, on
    elif isfile(fp):
        table = id2file_from_map(fp)
        if table:
            if wanted:
                lookup = dict(table)
                if any(x not in lookup for x in wanted):
                    raise ValueError(INCONSISTENT)
                files = {lookup[x]: x for x in wanted}
            else:
                wanted = [sample for sample, _ in table]
                files = {path: sample for sample, path in table}
            click.echo(f'Number of alignment files to read: {len(files)}.')
            res = wanted, files, bool(demux)
        else:
            res = single(fp, path2stem(fp, ext), f'Input alignment file: {fp}.')
    else:
        raise ValueError(f'"{fp}" is not a valid file or directory.')
    click.echo(f'Demultiplexing: {"on" if res[2] else "off"}.')
    return res


def parse_exclude(exclude=None):
    """Subjects to exclude: comma list or id file (workflow.py:483-503)."""
    if not exclude:
        return None
    ids = _id_list(exclude)
    click.echo(f'Number of subjects to exclude: {len(ids)}.')
    return set(ids)


def parse_strata(fp=None, samples=None):
    """{sample: stratification file} (workflow.py:506-533)."""
    if not fp:
        return None
    click.echo(f'Stratification file directory: {fp}.')
    found = id2file_from_dir(fp, ids=samples)
    if samples and len(samples) > len(found):
        raise ValueError(
            'Cannot locate stratification files for one or more samples.')
    return {sample: join(fp, name) for sample, name in found.items()}


# The gene coordinates are read on a thread while the hierarchy is (`workflow`
# starts it; the parser is native and holds no interpreter lock): build_mapper
# picks the table up -- or the error, which surfaces where it always did.
_coords_ahead = {}


def start_coords_ahead(coords_fp, zippers=None):
    import threading
    if _coords_ahead or not coords_fp:
        return
    box = {'fp': coords_fp}

    def work():
        try:
            import time
            t0 = time.perf_counter()
            table = load_gene_coords_file(coords_fp, zippers)
            t1 = time.perf_counter()
            table.names
            box['table'] = table
            box['laps'] = (t0, t1, time.perf_counter())
        except BaseException as e:     # noqa: BLE001 - raised by build_mapper
            box['err'] = e
    th = threading.Thread(target=work, name='wk-coords', daemon=True)
    _coords_ahead['x'] = (th, box)
    th.start()


def _coords_table(coords_fp, zippers):
    th, box = _coords_ahead.pop('x', (None, None))
    if th is not None:
        import time
        t_join = time.perf_counter()
        th.join()
        if os.environ.get('WOLTKA_DTOK_TIMING') and 'laps' in box:
            import sys
            t0, t1, t2 = box['laps']
            print('[coords] reader ahead: read + parse %.3f s, names %.3f s; '
                  'begun %.3f s before it was asked for, waited for %.3f s'
                  % (t1 - t0, t2 - t1, t_join - t0,
                     time.perf_counter() - t_join), file=sys.stderr)
        if box['fp'] == coords_fp:
            if 'err' in box:
                raise box['err']
            return box['table']
    table = load_gene_coords_file(coords_fp, zippers)
    table.names     # (as Python strings now, while the device context opens on its thread)
    return table


def build_mapper(coords_fp=None, outcov_dir=None, overlap=None, chunk=None,
                 zippers=None):
    """Plain mapper, or coord-match mapper when gene coordinates are given
    (workflow.py:536-585).  Returns (mapper, chunk); ``chunk`` stays ``None``
    unless the user set it (the device default is chosen in ``classify``)."""
    if coords_fp:
        click.echo('Reading gene coordinates...', nl=False)
        table = _coords_table(coords_fp, zippers)
        click.echo(' Done.')
        click.echo(f'  Total number of host sequences: {len(table)}.')
        return OrdinalMapper(table, th=overlap and overlap / 100), chunk
    return (range_mapper if outcov_dir else plain_mapper), chunk


def parse_sizes(sizes, mapper, zippers=None):
    """Feature sizes for ``--sizes`` as reciprocals (workflow.py:588-633):
    a two-column map file, or "." = gene lengths from the coordinates."""
    if not sizes:
        return None
    from_coords = sizes == '.'
    click.echo('Calculating gene lengths from coordinates...' if from_coords
               else f'Reading subject sizes file: {basename(sizes)}...',
               nl=False)
    if from_coords:
        if not isinstance(mapper, OrdinalMapper):
            raise ValueError('Gene coordinates file is not provided.')
        lengths = mapper.table.gene_lengths(mapper.prefix).items()
    else:
        with readzip(sizes, zippers) as fh:
            lengths = [(k, float(v)) for k, v in read_map_1st(fh)]
    click.echo(' Done.')
    return {name: 1 / size for name, size in lengths}


def prepare_ranks(ranks=None, outmap_dir=None, tree=None, rankdic=None):
    """Rank list and read-map directories (workflow.py:636-695): the ranks
    must exist in the classification system ("none" / "free" always do); no
    rank given means free-rank classification if there is a hierarchy, plain
    subject counting otherwise.  Several ranks get one map directory each."""
    chosen = ranks.split(',') if ranks else ['free' if tree else 'none']
    if ranks and rankdic is not None:
        native = getattr(rankdic, 'native', None)
        known = (set(rankdic.values()) if native is None
                 else set(native.ranks_in_use)) | {'none', 'free'}
        unknown = sorted(set(chosen) - known)
        if unknown:
            raise ValueError(f'Ranks {", ".join(unknown)} are not found in '
                             'classification system.')
    click.echo(f'Classification will operate on these ranks: '
               f'{", ".join(chosen)}.')
    if not outmap_dir:
        return chosen, None
    makedirs(outmap_dir, exist_ok=True)
    click.echo(f'Read-to-feature maps will be saved to: {outmap_dir}.')
    if len(chosen) == 1:
        return chosen, {chosen[0]: outmap_dir}
    dirs = {rank: join(outmap_dir, rank) for rank in chosen}
    for path in dirs.values():
        makedirs(path, exist_ok=True)
    return chosen, dirs


def build_hierarchy(names_fps=[], nodes_fps=[], newick_fps=[], lineage_fps=[],
                    columns_fps=[], map_fps=[], map_rank=None, zippers=None):
    """Read all hierarchy files into (tree, rankdic, namedic, root)
    (workflow.py:698-815).  The three dicts are views of one native table
    (``hierarchy.NativeTaxonomy``): nodes / names / map files are parsed by the
    native ingest on all threads (csrc/wk_hierarchy.cpp), the formats with
    few lines (Newick, lineage strings, rank columns) and text the native
    readers refuse go through the Python readers of ``tree.py`` and are
    merged into the same table; merging (``util.update_dict``) and
    ``fill_root`` happen natively."""
    from . import _native as nat
    from .classify import tokenizer_threads
    from .hierarchy import NativeTaxonomy
    tax = NativeTaxonomy(tokenizer_threads())
    is_build = any([names_fps, nodes_fps, newick_fps, lineage_fps,
                    columns_fps, map_fps])
    if is_build:
        click.echo('Constructing classification system...')

    streamed = {}   # bytes of inputs that can be read only once (FIFOs ...)

    def python_read(fp, reader):
        if fp in streamed:
            import io
            return reader(io.TextIOWrapper(io.BytesIO(streamed.pop(fp))))
        with readzip(fp, zippers) as f:
            return reader(f)

    def native_read(fp, kind, rank=None):
        """True when the native reader took the file."""
        try:
            with _file_bytes(fp, zippers) as buf:
                if isinstance(buf, bytes) and not os.path.isfile(fp):
                    streamed[fp] = buf
                tax.add_text(kind, buf, rank)
            streamed.pop(fp, None)
            return True
        except nat.HierarchyBuilder.Refused:
            return False

    def each(fps, label):
        for fp in fps:
            click.echo(f'  Parsing {label}: {basename(fp)}...', nl=False)
            yield fp
            click.echo(' Done.')

    for fp in each(names_fps, 'taxon names file'):
        if not native_read(fp, nat.HIER_NAMES):
            tax.update(nat.HIER_NAME, python_read(fp, read_names))
    for fp in each(nodes_fps, 'taxon nodes file'):
        if not native_read(fp, nat.HIER_NODES):
            tree_, rankdic_ = python_read(fp, read_nodes)
            tax.update(nat.HIER_PARENT, tree_)
            tax.update(nat.HIER_RANK, rankdic_)
    for fp in each(newick_fps, 'Newick tree file'):
        tax.update(nat.HIER_PARENT, python_read(fp, read_newick))
    for fp in each(lineage_fps, 'lineage file'):
        tree_, rankdic_ = python_read(fp, read_lineage)
        tax.update(nat.HIER_PARENT, tree_)
        tax.update(nat.HIER_RANK, rankdic_)
    for fp in each(columns_fps, 'columns file'):
        tree_, rankdic_ = python_read(fp, read_columns)
        tax.update(nat.HIER_PARENT, tree_)
        tax.update(nat.HIER_RANK, rankdic_)
    if map_rank is None:
        map_rank = bool(map_fps) and not any([
            nodes_fps, newick_fps, lineage_fps, columns_fps])
    if map_rank:
        click.echo('  Will extract rank name from map filename.')
    for fp in map_fps:
        click.echo(f'  Parsing simple map file: {basename(fp)}...', nl=False)
        rank = stem2rank(path2stem(fp)) if map_rank else None
        if not native_read(fp, nat.HIER_MAP, rank):
            map_ = dict(python_read(fp, read_map_1st))
            tax.update(nat.HIER_PARENT, map_)
            if map_rank:
                tax.update(nat.HIER_RANK, {k: rank for k in set(map_.values())})
        click.echo(' Done.')
    tax.finish()
    if is_build:
        click.echo('Classification system constructed.')
        click.echo(f'  Total number of classification units: {tax.n_nodes}.')
    return tax.tree, tax.rankdic, tax.namedic, tax.root


@contextlib.contextmanager
def _file_bytes(fp, zippers=None):
    """The bytes of a (possibly compressed) file as a buffer: a plain file is
    memory-mapped, a compressed one inflated into memory."""
    import mmap
    from os.path import splitext
    from .file import ZIP_BY_EXT
    if ZIP_BY_EXT.get(splitext(fp)[1]) is None:
        import stat
        with open(fp, 'rb') as f:
            st = os.fstat(f.fileno())
            mm = None
            # only a regular file with a size can be mapped; a FIFO, a process
            # substitution, /dev/stdin or a procfs file reports size 0 and is
            # streamed instead
            if stat.S_ISREG(st.st_mode) and st.st_size > 0:
                try:
                    mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
                except (OSError, ValueError):
                    mm = None
            if mm is None:
                yield f.read()
                return
            view = memoryview(mm)
            try:
                yield view
            finally:
                view.release()
                mm.close()
    else:
        with readzip_bytes(fp, zippers) as f:
            yield f.read()


def read_strata(strata_fp, zippers=None):
    """{query: stratum} of one sample (workflow.py:912-938)."""
    with readzip(strata_fp, zippers) as fh:
        pairs = dict(read_map_uniq(fh))
    if pairs:
        return pairs
    raise ValueError('No stratification information is found in file: '
                     f'{basename(strata_fp)}.')


def scale_factor(s):
    """"1k" / "1M" / plain numbers, case-insensitive; integral values stay
    ``int`` (woltka/util.py:97-127)."""
    s = s.strip().lower()
    mult = 1
    if s.endswith('k'):
        mult, s = 1000, s[:-1]
    elif s.endswith('m'):
        mult, s = 1000000, s[:-1]
    try:
        return int(s) * mult
    except ValueError:
        try:
            return float(s) * mult
        except ValueError:
            raise ValueError('Invalid scale factor.')


def frac_profiles(data, frac=False):
    """Divide by the per-sample total (workflow.py:1061-1084)."""
    if frac:
        for profile in data.values():
            for cells in profile.values():
                total = sum(cells.values())
                if total:
                    for feature, value in cells.items():
                        cells[feature] = value / total


def scale_profiles(data, scale=None):
    """Multiply by a factor (workflow.py:1087-1103)."""
    if not scale:
        return
    factor = scale_factor(scale)
    for profile in data.values():
        for sample in profile.values():
            for feature in sample:
                sample[feature] *= factor


def round_half_snap(value, digits=None):
    """One cell of util.round_dict (woltka/util.py:342-348)."""
    error = 1e-7 / 10 ** digits if digits else 1e-7
    near = round(value * 2, digits) / 2
    if abs(value - near) <= error:
        return round(near, digits)
    return round(value, digits)


def _round_bulk(sample):
    """`round_half_snap` (no digits) over a whole sample in numpy; False when
    the cells are not plain numbers below 2^52.  Binary64 arithmetic is the
    same in both (``* 2`` and ``/ 2`` are exact, ``round`` of a float and
    ``rint`` both round half to even)."""
    import numpy as np
    vals = list(sample.values())
    if not set(map(type, vals)) <= {int, float}:
        return False
    try:
        v = np.array(vals, dtype=np.float64)
    except (OverflowError, ValueError):
        return False
    if not np.all(np.abs(v) < 2.0 ** 52):   # (also False for nan / inf)
        return False
    near = np.rint(v * 2) / 2
    r = np.where(np.abs(v - near) <= 1e-7, np.rint(near), np.rint(v))
    r = r.astype(np.int64)
    keep = r != 0
    keys = list(sample)
    sample.clear()
    if keep.all():
        sample.update(zip(keys, r.tolist()))
    else:
        from itertools import compress
        sample.update(zip(compress(keys, keep.tolist()), r[keep].tolist()))
    return True


def round_profiles(data, digits=None):
    """Round cells, drop zeros (workflow.py:1106-1119, util.round_dict).  An
    ``int`` cell rounds to itself (``round(v * 2) / 2 == v``), so only the
    other cells go through the rule."""
    from .cells import LazyCells
    for profile in data.values():
        for sample in profile.values():
            if type(sample) is LazyCells and sample.pending and \
                    digits is None and sample.round_bulk():
                continue        # (rounded as arrays)
            if digits is None and len(sample) > 256 and _round_bulk(sample):
                continue
            dead = []
            for feature, value in sample.items():
                if type(value) is int and digits is None:
                    if not value:
                        dead.append(feature)
                    continue
                r = round_half_snap(value, digits)
                if r:
                    sample[feature] = r
                else:
                    dead.append(feature)
            for feature in dead:
                del sample[feature]


def _write_one_sample(profile, columns, path):
    """A TSV table of one sample with plain feature ids and integer cells —
    the usual output of a run on one file — sorted and formatted natively
    (`wk_table_body`): what `prep_table` + `write_tsv` write, without a Python
    object per row.  Returns the number of features, or None when the table is
    not of that kind (the general writer then)."""
    import locale
    import numpy as np
    from . import _native as nat
    cols = [s for s in columns if s in profile]
    if len(cols) != 1:
        return None
    sample = profile[cols[0]]
    n = len(sample)
    if n < 1024 or not set(map(type, sample)) <= {str} or \
            not set(map(type, sample.values())) <= {int} or \
            locale.getpreferredencoding(False).lower().replace('-', '') != 'utf8':
        return None
    try:
        values = np.fromiter(sample.values(), np.int64, n)
        keys = '\n'.join(sample).encode()
        head = f'#FeatureID\t{cols[0]}\n'.encode()
    except (OverflowError, UnicodeEncodeError):
        return None
    if keys.count(b'\n') != n - 1:      # (a feature id with a line break)
        return None
    res = nat.table_body(keys, values)
    if res is None:
        return None
    with openzip(path, 'wb') as fh:
        fh.write(head)
        fh.write(res[0])
    return res[1]


def write_profiles(data, fp, is_biom=None, samples=None, tree=None,
                   rankdic=None, namedic=None, name_as_id=False,
                   add_rank=False, add_lineage=False):
    """Write one table per rank (workflow.py:1122-1205).  One rank: ``fp`` is
    the file and its extension picks the format unless --to-biom / --to-tsv
    said so; several ranks: ``fp`` is a directory of ``<rank>.biom`` (default)
    or ``<rank>.tsv`` files."""
    if not fp:
        return
    ranks = sorted(data)
    if len(ranks) > 1:
        makedirs(fp, exist_ok=True)
        biom = is_biom is not False
        targets = [(r, join(fp, f"{r}.{'biom' if biom else 'tsv'}"))
                   for r in ranks]
    else:
        biom = fp.endswith('.biom') if is_biom is None else is_biom
        targets = [(ranks[0], fp)]
    label = 'BIOM' if biom else 'TSV'
    click.echo(f'Format of output feature table(s): {label}.')
    click.echo(f'Writing output profiles in {label} format...')
    columns = samples or sorted(allkeys(data))
    from .cells import write_lazy_table
    for rank, path in targets:
        if not (biom or add_lineage or add_rank or namedic):
            done = write_lazy_table(data[rank], columns, path, openzip)
            if done is not None:
                click.echo(f'  Rank: {rank}, samples: {done[0]}, features: '
                           f'{done[1]}.')
                continue
            rows = _write_one_sample(data[rank], columns, path)
            if rows is not None:
                click.echo(f'  Rank: {rank}, samples: 1, features: {rows}.')
                continue
        table = prep_table(data[rank], columns,
                           tree if add_lineage else None,
                           rankdic if add_rank else None, namedic,
                           name_as_id and namedic is not None)
        write_table(table, path, biom)
        click.echo(f'  Rank: {rank}, samples: {len(table[2])}, features: '
                   f'{len(table[1])}.')
    click.echo('Profiles written.')
