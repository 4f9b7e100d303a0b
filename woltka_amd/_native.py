"""ctypes binding of ``libwoltka_hip.so`` (C ABI in ``include/woltka_hip.h``).

This is the only route from the Python host layer to the device.  There is no
CPU fallback: if the shared library is missing, or no HIP device is usable,
the functions here raise ``RuntimeError`` — they never compute on the host.
"""
import ctypes as C
import os
from fractions import Fraction

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# WOLTKA_HIP_LIB lets tools/ load a measurement build (e.g. -DWK_ABLATE)
LIB_PATH = os.environ.get('WOLTKA_HIP_LIB') or os.path.join(
    _HERE, 'libwoltka_hip.so')

# constants mirrored from include/woltka_hip.h
ABI_VERSION = 1
OK, E_HIP, E_ARG, E_STATE, E_CAPACITY, E_RANGE, E_TABLE_FULL = (
    0, -1, -2, -3, -4, -5, -6)
KEY_FEATURE_BITS, KEY_GROUP_BITS, KEY_K_BITS, KEY_JOB_BITS = 28, 21, 12, 3
MAX_JOBS, MAX_K = 8, 4095
FEATURE_UNASSIGNED, MAX_FEATURE = 0x0FFFFFFF, 0x0FFFFFFE
MODE_NONE, MODE_FREE, MODE_RANK = 0, 1, 2
F_UNIQ, F_ABOVE, F_SUBOK, F_UNASSIGNED, F_SIZED = 1, 2, 4, 8, 16
ASSIGN_NONE, ASSIGN_MULTI, ASSIGN_EMPTY = -1, -2, -3
SUBJ_IS_SET, SUBJ_INDEXED, GROUP_UNIFORM = 1, 2, 4
MAX_RANK_SLOTS = MAX_JOBS * 4

# every symbol the header declares (checked by tests/test_abi.py)
SYMBOLS = (
    'wk_abi_version', 'wk_build_id', 'wk_device_count', 'wk_device_pci_bus_id',
    'wk_create', 'wk_destroy',
    'wk_last_error',
    'wk_device_name', 'wk_sync', 'wk_set_option', 'wk_tune', 'wk_set_tree',
    'wk_build_rank_table', 'wk_get_rank_table', 'wk_set_genes',
    'wk_set_subjects',
    'wk_counts_reserve', 'wk_counts_clear', 'wk_counts_fetch',
    'wk_log_reserve', 'wk_log_fetch',
    'wk_chunk_stage', 'wk_classify_staged', 'wk_classify_chunk',
    'wk_ordinal_stage', 'wk_ordinal_match', 'wk_ordinal_count',
    'wk_set_uniform_group', 'wk_chunk_download', 'wk_ordinal_hit_offsets',
    'wk_ordinal_pair_genes',
    'wk_blob_join', 'wk_table_body', 'wk_table_rows', 'wk_host_alloc', 'wk_host_free', 'wk_host_register', 'wk_host_unregister',
    'wk_words_begin', 'wk_words_append',
    'wk_words_wait', 'wk_words_flush', 'wk_words_pending',
    'wk_get_stats', 'wk_reset_stats', 'wk_timer_begin', 'wk_timer_end',
    'wk_timer_ms', 'wk_profile_kernels', 'wk_last_kernel_ms',
    'wk_tok_create', 'wk_tok_destroy', 'wk_tok_last_error',
    'wk_tok_set_exclude', 'wk_tok_sam_tail', 'wk_tok_sam', 'wk_tok_text',
    'wk_tok_boundary',
    'wk_tok_fetch', 'wk_tok_fetch_packed', 'wk_tok_set_subject_map',
    'wk_tok_read', 'wk_tok_trim', 'wk_tok_sam_span', 'wk_tok_span', 'wk_tok_set_header_state',
    'wk_dtok_format',
    'wk_dtok_copy', 'wk_dtok_copy_ahead', 'wk_dtok_copy_wait',
    'wk_dtok_copy_drop', 'wk_dtok_subject_map', 'wk_dtok_ahead_room', 'wk_dtok_text_back', 'wk_dtok_expect', 'wk_dtok_scan', 'wk_dtok_emit', 'wk_dtok_stage_hits', 'wk_dtok_stage_hits_append',
    'wk_dtok_scan_emit', 'wk_dtok_scan_emit_begin', 'wk_dtok_scan_emit_end',
    'wk_dtok_keep_reads', 'wk_readmap_tables',
    'wk_dtok_readmap',
    'wk_dtok_readmap_fetch', 'wk_strata_load', 'wk_strata_labels',
    'wk_strata_groups', 'wk_strata_clear',
    'wk_tok_subjects',
    'wk_tok_new_subjects', 'wk_tok_fetch_groups', 'wk_tok_strata_clear',
    'wk_tok_strata_load', 'wk_tok_strata_labels', 'wk_tok_strata_select',
    'wk_tok_strata_swap', 'wk_format_readmap',
    'wk_tok_fetch_samples', 'wk_tok_new_samples', 'wk_preorder',
    'wk_hier_create', 'wk_hier_destroy', 'wk_hier_last_error',
    'wk_hier_add_text', 'wk_hier_update', 'wk_hier_finish', 'wk_hier_arrays',
    'wk_hier_root', 'wk_hier_lookup', 'wk_hier_node_names', 'wk_hier_get',
    'wk_hier_size', 'wk_hier_keys', 'wk_hier_ranks',
    'wk_coords_parse', 'wk_coords_error', 'wk_coords_sizes',
    'wk_coords_fetch', 'wk_coords_free',
    'wk_gz_bound', 'wk_gz_member', 'wk_crc32', 'wk_gz_inflate_members',
    'wk_gunzip_open', 'wk_gunzip_read', 'wk_gunzip_error', 'wk_gunzip_close',
    'wk_text_upload', 'wk_text_clear', 'wk_h2d_rate',
    'wk_dtok_fused_counts', 'wk_ordinal_chunk_counts')


class Job(C.Structure):
    """``struct wk_job``."""
    _fields_ = [('mode', C.c_int32), ('rank_slot', C.c_int32),
                ('flags', C.c_uint32), ('_pad', C.c_uint32),
                ('major', C.c_double)]


class Stats(C.Structure):
    """``struct wk_stats``."""
    _fields_ = [('n_reads', C.c_int64), ('n_records', C.c_int64),
                ('n_pairs', C.c_int64), ('table_used', C.c_int64)]


_lib = None


def load_library():
    """Load the in-tree shared library and declare its prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise RuntimeError(
            f'{LIB_PATH} is missing: build it with `python -c "import '
            '__graft_entry__ as g; g.build()"` (hipcc --offload-arch=gfx950). '
            'There is no CPU fallback.')
    lib = C.CDLL(LIB_PATH)
    p = C.c_void_p
    i32p, i64p, u32p, u64p = (C.POINTER(C.c_int32), C.POINTER(C.c_int64),
                              C.POINTER(C.c_uint32), C.POINTER(C.c_uint64))
    proto = {
        'wk_abi_version': (C.c_int, []),
        'wk_build_id': (C.c_char_p, []),
        'wk_device_count': (C.c_int, []),
        'wk_device_pci_bus_id': (C.c_int, [C.c_int, C.c_char_p, C.c_size_t]),
        'wk_create': (C.c_int, [C.c_int, C.POINTER(p)]),
        'wk_destroy': (None, [p]),
        'wk_last_error': (C.c_char_p, [p]),
        'wk_device_name': (C.c_int, [p, C.c_char_p, C.c_size_t]),
        'wk_sync': (C.c_int, [p]),
        'wk_set_option': (C.c_int, [p, C.c_char_p, C.c_int64]),
        'wk_tune': (C.c_int, [p, C.c_char_p, C.c_int64]),
        'wk_set_tree': (C.c_int, [p, i32p, i32p, i32p, C.c_int32]),
        'wk_build_rank_table': (C.c_int, [p, C.c_int32, C.c_int32]),
        'wk_get_rank_table': (C.c_int, [p, C.c_int32, i32p]),
        'wk_set_genes': (C.c_int, [p, i32p, C.c_int32, i32p, i32p, i32p,
                                   C.c_int32]),
        'wk_set_subjects': (C.c_int, [p, i32p, C.c_int32]),
        'wk_counts_reserve': (C.c_int, [p, C.c_int64]),
        'wk_counts_clear': (C.c_int, [p]),
        'wk_counts_fetch': (C.c_int, [p, u64p, i64p, C.c_int64, i64p]),
        'wk_log_reserve': (C.c_int, [p, C.c_int64]),
        'wk_log_fetch': (C.c_int, [p, i32p, C.c_int64, i64p]),
        'wk_chunk_stage': (C.c_int, [p, i32p, i32p, C.c_int64, i32p,
                                     C.c_int]),
        'wk_classify_staged': (C.c_int, [p, C.POINTER(Job), C.c_int32, i32p]),
        'wk_classify_chunk': (C.c_int, [p, C.POINTER(Job), C.c_int32, i32p,
                                        i32p, C.c_int64, i32p, C.c_int,
                                        i32p]),
        'wk_ordinal_stage': (C.c_int, [p, i32p, i32p, i32p, u32p, C.c_int64,
                                       i32p, C.c_int64, i32p, C.c_double]),
        'wk_ordinal_match': (C.c_int, [p]),
        'wk_ordinal_count': (C.c_int, [p, C.POINTER(Job), C.c_int32]),
        'wk_set_uniform_group': (C.c_int, [p, C.c_int32]),
        'wk_ordinal_hit_offsets': (C.c_int, [p, i32p, C.c_int64]),
        'wk_chunk_download': (C.c_int, [p, i32p, C.c_int64, i32p, C.c_int64,
                                        i64p, i64p]),
        'wk_host_alloc': (C.c_int, [p, C.c_size_t, C.POINTER(C.c_void_p)]),
        'wk_host_free': (C.c_int, [p, C.c_void_p]),
        'wk_blob_join': (C.c_int, [C.c_char_p, i64p, C.c_int64, C.c_char,
                                   C.c_void_p]),
        'wk_table_body': (C.c_int, [C.c_char_p, C.c_int64, i64p, C.c_int64,
                                    C.c_int, C.c_void_p, C.c_int64, i64p, i64p]),
        'wk_host_register': (C.c_int, [p, C.c_void_p, C.c_size_t]),
        'wk_host_unregister': (C.c_int, [p, C.c_void_p]),
        'wk_words_begin': (C.c_int, [p, C.POINTER(Job), C.c_int32, C.c_int32,
                                     C.POINTER(C.c_int)]),
        'wk_words_append': (C.c_int, [p, u32p, C.c_int64, C.c_int64, C.c_int]),
        'wk_words_wait': (C.c_int, [p, C.c_int]),
        'wk_words_flush': (C.c_int, [p]),
        'wk_words_pending': (C.c_int, [p, i64p, i64p]),
        'wk_get_stats': (C.c_int, [p, C.POINTER(Stats)]),
        'wk_reset_stats': (C.c_int, [p]),
        'wk_timer_begin': (C.c_int, [p]),
        'wk_timer_end': (C.c_int, [p]),
        'wk_timer_ms': (C.c_int, [p, C.POINTER(C.c_double)]),
        'wk_profile_kernels': (C.c_int, [p, C.c_int]),
        'wk_last_kernel_ms': (C.c_int, [p, C.c_char_p,
                                        C.POINTER(C.c_double)]),
        'wk_preorder': (C.c_int, [i64p, C.c_int64, C.c_int64, i64p, i64p, i64p,
                                  i64p]),
        'wk_tok_create': (C.c_int, [C.c_int, C.POINTER(p)]),
        'wk_tok_destroy': (None, [p]),
        'wk_tok_last_error': (C.c_char_p, [p]),
        'wk_tok_set_exclude': (C.c_int, [p, C.c_char_p, i32p, C.c_int32]),
        'wk_tok_sam_tail': (C.c_int, [p, C.c_char_p, C.c_int64,
                                      C.POINTER(C.c_int64)]),
        'wk_tok_sam': (C.c_int, [p, C.c_void_p, C.c_int64, C.c_int, C.c_int,
                                 C.c_int, C.c_int, i64p, i64p, i64p]),
        'wk_tok_boundary': (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_int64,
                                      C.c_int64, i64p]),
        'wk_tok_text': (C.c_int, [p, C.c_int, C.c_void_p, C.c_int64, C.c_int,
                                  C.c_int, C.c_int, C.c_int, i64p, i64p,
                                  i64p]),
        'wk_tok_fetch': (C.c_int, [p, i32p, i32p, i32p, i32p, u32p, u64p]),
        'wk_tok_fetch_packed': (C.c_int, [p, u32p, i32p, u64p, i64p]),
        'wk_tok_set_subject_map': (C.c_int, [p, i32p, C.c_int32]),
        'wk_tok_read': (C.c_int, [p, C.c_int, C.c_int64, C.c_void_p, C.c_int64,
                                  i64p]),
        'wk_tok_trim': (C.c_int, [p, C.c_void_p, C.c_int, C.c_int64, C.c_int64,
                                  C.c_int64, C.c_int, C.c_void_p, C.c_int64,
                                  i64p, i64p]),
        'wk_tok_sam_span': (C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_int,
                                      i64p, i64p, C.POINTER(C.c_int)]),
        'wk_tok_span': (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_int64,
                                  C.c_int, C.c_int, i64p, i64p,
                                  C.POINTER(C.c_int)]),
        'wk_dtok_format': (C.c_int, [p, C.c_int]),
        'wk_tok_set_header_state': (C.c_int, [p, C.c_int]),
        'wk_dtok_copy': (C.c_int, [p, C.c_void_p, C.c_int64, C.c_int64]),
        'wk_dtok_copy_ahead': (C.c_int, [p, C.c_void_p, C.c_int64, C.c_int64,
                                         C.POINTER(C.c_int32)]),
        'wk_dtok_copy_wait': (C.c_int, [p, C.c_int32]),
        'wk_dtok_copy_drop': (C.c_int, [p]),
        'wk_dtok_subject_map': (C.c_int, [p, i32p, C.c_int32]),
        'wk_dtok_ahead_room': (C.c_int, [p, C.POINTER(C.c_int32)]),
        'wk_dtok_text_back': (C.c_int, [p, C.c_void_p, C.c_int64]),
        'wk_dtok_expect': (C.c_int, [p, C.c_int64]),
        'wk_dtok_scan': (C.c_int, [p, p, C.c_void_p, C.c_int64, C.c_int64,
                                   C.c_int, i64p, C.POINTER(C.c_int)]),
        'wk_dtok_stage_hits': (C.c_int, [p, i32p, C.c_int32, C.c_double, i64p,
                                         i64p, C.POINTER(C.c_int)]),
        'wk_dtok_stage_hits_append': (C.c_int, [p, i32p, C.c_int32, C.c_double,
                                                C.POINTER(Job), C.c_int32, i64p,
                                                i64p, C.POINTER(C.c_int),
                                                C.POINTER(C.c_int)]),
        'wk_dtok_emit': (C.c_int, [p, i64p, i64p, C.POINTER(C.c_int)]),
        'wk_dtok_scan_emit': (C.c_int, [p, p, C.c_void_p, C.c_int64, C.c_int64,
                                        i64p, C.POINTER(C.c_int),
                                        C.POINTER(C.c_int), i64p, i64p]),
        'wk_dtok_scan_emit_begin': (C.c_int, [p, p, C.c_void_p, C.c_int64,
                                              C.c_int64, C.POINTER(C.c_int)]),
        'wk_dtok_scan_emit_end': (C.c_int, [p, i64p, C.POINTER(C.c_int), i64p,
                                            i64p]),
        'wk_dtok_keep_reads': (C.c_int, [p, C.c_int]),
        'wk_readmap_tables': (C.c_int, [p, C.c_int32, i32p, C.c_int32, i32p,
                                        u32p, C.c_char_p, C.c_int32]),
        'wk_dtok_readmap': (C.c_int, [p, C.c_int32, i64p]),
        'wk_dtok_readmap_fetch': (C.c_int, [p, C.c_void_p, C.c_int64]),
        'wk_strata_load': (C.c_int, [p, C.c_void_p, C.c_int64, i64p, i32p,
                                     C.POINTER(C.c_int)]),
        'wk_strata_labels': (C.c_int, [p, i32p, i64p, i32p, C.c_int32, i32p]),
        'wk_strata_groups': (C.c_int, [p, i32p, i32p, C.c_int32]),
        'wk_strata_clear': (C.c_int, [p]),
        'wk_tok_subjects': (C.c_int, [p, i32p, i32p, i64p]),
        'wk_tok_new_subjects': (C.c_int, [p, C.c_char_p, i32p]),
        'wk_tok_fetch_groups': (C.c_int, [p, i32p]),
        'wk_tok_strata_clear': (C.c_int, [p]),
        'wk_tok_fetch_samples': (C.c_int, [p, i32p]),
        'wk_tok_new_samples': (C.c_int, [p, C.c_char_p, i64p, i32p]),
        'wk_tok_strata_load': (C.c_int, [p, C.c_void_p, C.c_int64, i64p, i32p]),
        'wk_tok_strata_labels': (C.c_int, [p, C.c_char_p, i64p]),
        'wk_tok_strata_select': (C.c_int, [p, C.c_int]),
        'wk_tok_strata_swap': (C.c_int, [p]),
        'wk_format_readmap': (C.c_int, [C.c_void_p, u64p, i32p, C.c_int64, i64p,
                                        i32p, i32p, C.c_char_p, i64p, C.c_int32,
                                        C.c_int, C.c_int, C.c_void_p, C.c_int64,
                                        i64p]),
        'wk_table_rows': (C.c_int, [C.c_char_p, C.c_int64, C.c_int32, C.c_char_p,
                                    C.c_int64, C.c_int32, i32p, i32p, i64p,
                                    C.c_int64, C.c_int32, C.c_void_p,
                                    C.c_int64, i64p, i64p]),
        'wk_ordinal_pair_genes': (C.c_int, [p, i32p, C.c_int64]),
        'wk_hier_create': (C.c_int, [C.c_int, C.POINTER(p)]),
        'wk_hier_destroy': (None, [p]),
        'wk_hier_last_error': (C.c_char_p, [p]),
        'wk_hier_add_text': (C.c_int, [p, C.c_int, C.c_void_p, C.c_int64,
                                       C.c_char_p]),
        'wk_hier_update': (C.c_int, [p, C.c_int, C.c_char_p, i64p, C.c_char_p,
                                     i64p, C.POINTER(C.c_uint8), C.c_int64]),
        'wk_hier_finish': (C.c_int, [p, i64p, i32p]),
        'wk_hier_arrays': (C.c_int, [p, i32p, i32p, i32p, i32p]),
        'wk_hier_root': (C.c_int, [p, i32p]),
        'wk_hier_lookup': (C.c_int, [p, C.c_char_p, i64p, C.c_int64, i32p]),
        'wk_hier_node_names': (C.c_int, [p, i32p, C.c_int64, C.c_void_p,
                                         C.c_int64, i64p]),
        'wk_hier_get': (C.c_int, [p, C.c_int, C.c_char_p, C.c_int64,
                                  C.c_void_p, C.c_int64, i64p]),
        'wk_hier_size': (C.c_int64, [p, C.c_int]),
        'wk_hier_keys': (C.c_int, [p, C.c_int, C.c_void_p, C.c_int64, i64p]),
        'wk_hier_ranks': (C.c_int, [p, C.c_void_p, C.c_int64, i64p, i64p]),
        'wk_coords_parse': (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(p)]),
        'wk_coords_error': (C.c_char_p, [p]),
        'wk_coords_sizes': (C.c_int, [p, i32p, i32p, i64p, i64p,
                                      C.POINTER(C.c_int)]),
        'wk_coords_fetch': (C.c_int, [p, i32p, i32p, i32p, i32p, C.c_void_p,
                                      i64p, C.c_void_p, i64p]),
        'wk_coords_free': (None, [p]),
        'wk_gz_bound': (C.c_int64, [C.c_int64]),
        'wk_gz_member': (C.c_int64, [C.c_void_p, C.c_int64, C.c_void_p,
                                     C.c_int64]),
        'wk_crc32': (C.c_uint32, [C.c_uint32, C.c_void_p, C.c_int64]),
        'wk_gz_inflate_members': (C.c_int64, [C.c_void_p, i64p, i64p, C.c_int64,
                                              C.c_void_p, i64p, C.c_int]),
        'wk_gunzip_open': (p, [C.c_char_p, C.c_int, C.c_char_p, C.c_size_t]),
        'wk_gunzip_read': (C.c_int64, [p, C.c_void_p, C.c_int64]),
        'wk_gunzip_error': (C.c_char_p, [p]),
        'wk_gunzip_close': (None, [p]),
        'wk_text_upload': (C.c_int, [p, C.c_void_p, C.c_int64, C.c_int64]),
        'wk_text_clear': (C.c_int, [p]),
        'wk_dtok_fused_counts': (C.c_int, [p, i64p, i64p]),
        'wk_ordinal_chunk_counts': (C.c_int, [p, i64p, i64p]),
        'wk_h2d_rate': (C.c_int, [p, C.c_int64, C.c_int,
                                  C.POINTER(C.c_double)]),
    }
    for name, (res, args) in proto.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.wk_abi_version() != ABI_VERSION:
        raise RuntimeError('libwoltka_hip.so ABI version mismatch')
    _lib = lib
    return lib


def _arr(a, dtype):
    """Contiguous array of the given dtype (no copy when already so)."""
    return np.ascontiguousarray(a, dtype=dtype)


def _ptr(a, ctype):
    return None if a is None else a.ctypes.data_as(C.POINTER(ctype))


def device_pci_bus_id(device):
    """PCI address of HIP device ``device`` (e.g. '0000:c1:00.0')."""
    buf = C.create_string_buffer(64)
    if load_library().wk_device_pci_bus_id(int(device), buf, 64) != OK:
        raise RuntimeError('no such HIP device')
    return buf.value.decode()


def decode_keys(keys):
    """Split packed count keys into (job, k, group, feature) arrays."""
    keys = np.asarray(keys, dtype=np.uint64)
    feat = (keys & np.uint64((1 << KEY_FEATURE_BITS) - 1)).astype(np.int64)
    grp = ((keys >> np.uint64(KEY_FEATURE_BITS)) &
           np.uint64((1 << KEY_GROUP_BITS) - 1)).astype(np.int64)
    k = ((keys >> np.uint64(KEY_FEATURE_BITS + KEY_GROUP_BITS)) &
         np.uint64((1 << KEY_K_BITS) - 1)).astype(np.int64)
    job = (keys >> np.uint64(61)).astype(np.int64)
    return job, k, grp, feat


class Context:
    """One device context (one per GPU; use from one thread at a time)."""

    def __init__(self, device=0):
        self._lib = load_library()
        h = C.c_void_p()
        rc = self._lib.wk_create(int(device), C.byref(h))
        if rc != OK:
            msg = self._lib.wk_last_error(None).decode()
            raise RuntimeError(f'wk_create(device={device}) failed: {msg}')
        self._h = h
        self.device = device
        self.n_nodes = 0

    # -- plumbing ---------------------------------------------------------
    def _check(self, rc):
        if rc == OK:
            return
        msg = self._lib.wk_last_error(self._h).decode()
        if rc in (E_ARG, E_RANGE):
            raise ValueError(msg)
        if rc in (E_CAPACITY, E_TABLE_FULL):
            raise OverflowError(msg)
        raise RuntimeError(msg)

    def close(self):
        if getattr(self, '_h', None):
            # (a thread that pins buffers ahead, hostio.open_context_ahead)
            th = getattr(self, '_ring_thread', None)
            if th is not None:
                self._ring_stop = True
                th.join()
                self._ring_thread = None
            self._lib.wk_destroy(self._h)
            self._h = None

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
    def device_name(self):
        buf = C.create_string_buffer(256)
        self._check(self._lib.wk_device_name(self._h, buf, 256))
        return buf.value.decode()

    def sync(self):
        self._check(self._lib.wk_sync(self._h))

    def set_option(self, name, value):
        self._check(self._lib.wk_set_option(self._h, name.encode(),
                                            int(value)))

    # -- static state -----------------------------------------------------
    def tune(self, name, value):
        """A measurement knob (``wk_tune``: launch shapes, ablation switches;
        bench.py, tools/ and tests only — results never depend on them)."""
        self._check(self._lib.wk_tune(self._h, name.encode(), int(value)))

    def set_tree(self, parent, last, rank_code):
        parent, last, rank_code = (_arr(parent, np.int32),
                                   _arr(last, np.int32),
                                   _arr(rank_code, np.int32))
        n = parent.size
        if last.size != n or rank_code.size != n:
            raise ValueError('tree arrays differ in length')
        self._check(self._lib.wk_set_tree(
            self._h, _ptr(parent, C.c_int32), _ptr(last, C.c_int32),
            _ptr(rank_code, C.c_int32), n))
        self.n_nodes = n

    def build_rank_table(self, slot, rank_code):
        self._check(self._lib.wk_build_rank_table(self._h, slot, rank_code))

    def get_rank_table(self, slot):
        out = np.empty(self.n_nodes, dtype=np.int32)
        self._check(self._lib.wk_get_rank_table(self._h, slot,
                                                _ptr(out, C.c_int32)))
        return out

    def set_genes(self, genome_off, start0, end, gene_feature):
        genome_off = _arr(genome_off, np.int32)
        start0, end, gene_feature = (_arr(start0, np.int32),
                                     _arr(end, np.int32),
                                     _arr(gene_feature, np.int32))
        self._check(self._lib.wk_set_genes(
            self._h, _ptr(genome_off, C.c_int32), genome_off.size - 1,
            _ptr(start0, C.c_int32), _ptr(end, C.c_int32),
            _ptr(gene_feature, C.c_int32), start0.size))

    def set_subjects(self, feature_of_subject):
        f = _arr(feature_of_subject, np.int32)
        self._check(self._lib.wk_set_subjects(self._h, _ptr(f, C.c_int32),
                                              f.size))

    # -- counts -----------------------------------------------------------
    def counts_reserve(self, min_slots):
        self._check(self._lib.wk_counts_reserve(self._h, int(min_slots)))

    def counts_clear(self):
        self._check(self._lib.wk_counts_clear(self._h))

    def counts_fetch(self):
        """Return (keys uint64[], counts int64[]) of the device count table."""
        n = C.c_int64(0)
        rc = self._lib.wk_counts_fetch(self._h, None, None, 0, C.byref(n))
        if rc == OK and n.value == 0:
            return np.empty(0, np.uint64), np.empty(0, np.int64)
        if rc not in (OK, E_CAPACITY):
            self._check(rc)
        keys = np.empty(n.value, dtype=np.uint64)
        vals = np.empty(n.value, dtype=np.int64)
        self._check(self._lib.wk_counts_fetch(
            self._h, _ptr(keys, C.c_uint64), _ptr(vals, C.c_int64), n.value,
            C.byref(n)))
        return keys[:n.value], vals[:n.value]

    def log_reserve(self, n_entries):
        self._check(self._lib.wk_log_reserve(self._h, int(n_entries)))
        self._log_cap = int(n_entries)

    def log_fetch(self):
        """Entries of the contribution log as int32[n, 4] = (feature, subject,
        job << 16 | divisor, group); empties the log."""
        out = np.empty((self._log_cap, 4), dtype=np.int32)
        n = C.c_int64(0)
        self._check(self._lib.wk_log_fetch(self._h, _ptr(out, C.c_int32),
                                           self._log_cap, C.byref(n)))
        return out[:n.value]

    # -- classify ---------------------------------------------------------
    @staticmethod
    def _jobs(jobs):
        arr = (Job * len(jobs))()
        for i, j in enumerate(jobs):
            arr[i] = j
        return arr

    def chunk_stage(self, subj, qoff, group=None, subj_is_set=False,
                    indexed=False):
        """``group``: None (group 0), one int (every read belongs to that
        group: WK_GROUP_UNIFORM) or an int32 array with one entry per read."""
        subj, qoff = _arr(subj, np.int32), _arr(qoff, np.int32)
        n_reads = qoff.size - 1
        flags = (SUBJ_IS_SET if subj_is_set else 0) | \
            (SUBJ_INDEXED if indexed else 0)
        if group is not None and np.ndim(group) == 0:
            group = np.array([int(group)], dtype=np.int32)
            flags |= GROUP_UNIFORM
        elif group is not None:
            group = _arr(group, np.int32)
            if group.size != n_reads:
                raise ValueError('group must have one entry per read')
        self._check(self._lib.wk_chunk_stage(
            self._h, _ptr(subj, C.c_int32), _ptr(qoff, C.c_int32), n_reads,
            _ptr(group, C.c_int32), flags))
        self._n_reads = n_reads

    def classify_staged(self, jobs, want_assign=False):
        out = None
        if want_assign:
            out = np.empty((len(jobs), self._n_reads), dtype=np.int32)
        self._check(self._lib.wk_classify_staged(
            self._h, self._jobs(jobs), len(jobs), _ptr(out, C.c_int32)))
        return out

    def classify_chunk(self, jobs, subj, qoff, group=None, subj_is_set=False,
                       want_assign=False, indexed=False):
        self.chunk_stage(subj, qoff, group, subj_is_set, indexed)
        return self.classify_staged(jobs, want_assign)

    # -- ordinal ----------------------------------------------------------
    def ordinal_stage(self, genome, beg, end, length, hoff, th, group=None):
        genome, beg, end = (_arr(genome, np.int32), _arr(beg, np.int32),
                            _arr(end, np.int32))
        length, hoff = _arr(length, np.uint32), _arr(hoff, np.int32)
        n_hits, n_reads = genome.size, hoff.size - 1
        if not (beg.size == end.size == length.size == n_hits):
            raise ValueError('hit arrays differ in length')
        if group is not None:
            group = _arr(group, np.int32)
        self._check(self._lib.wk_ordinal_stage(
            self._h, _ptr(genome, C.c_int32), _ptr(beg, C.c_int32),
            _ptr(end, C.c_int32), _ptr(length, C.c_uint32), n_hits,
            _ptr(hoff, C.c_int32), n_reads, _ptr(group, C.c_int32),
            float(th)))
        self._n_reads = n_reads

    def ordinal_match(self):
        self._check(self._lib.wk_ordinal_match(self._h))

    def ordinal_count(self, jobs):
        """Match the staged hits and count the reads' gene sets under every
        job (no per-read output)."""
        self._check(self._lib.wk_ordinal_count(self._h, self._jobs(jobs),
                                               len(jobs)))

    # -- packed records accumulated over a sample's chunks ------------------
    WORD_SUBJ_BITS, WORD_POS_SHIFT, WORD_SIZE_SHIFT = 23, 23, 27
    STAGE_SLOTS = 8

    def host_alloc(self, n, dtype=np.uint32):
        """A pinned host array of ``n`` elements (owned by the context; may
        be called from several threads: the host layer pins buffers ahead on
        a thread of its own, hostio.open_context_ahead)."""
        dt = np.dtype(dtype)
        out = C.c_void_p()
        self._check(self._lib.wk_host_alloc(self._h, int(n) * dt.itemsize,
                                            C.byref(out)))
        buf = (C.c_char * (int(n) * dt.itemsize)).from_address(out.value)
        arr = np.frombuffer(buf, dtype=dt, count=int(n))
        return arr

    def host_register(self, address, n):
        """Pin ``n`` bytes of the caller's (read-only, page-aligned) memory in
        place for asynchronous copies; False if the runtime refuses."""
        return self._lib.wk_host_register(self._h, C.c_void_p(int(address)),
                                          int(n)) == OK

    def host_unregister(self, address):
        self._check(self._lib.wk_host_unregister(self._h,
                                                 C.c_void_p(int(address))))

    def words_begin(self, jobs, group):
        ok = C.c_int(0)
        self._check(self._lib.wk_words_begin(self._h, self._jobs(jobs),
                                             len(jobs), int(group),
                                             C.byref(ok)))
        return bool(ok.value)

    def words_append(self, words, n_reads, slot=-1):
        words = _arr(words, np.uint32)
        self._check(self._lib.wk_words_append(
            self._h, _ptr(words, C.c_uint32), words.size, int(n_reads),
            int(slot)))

    def words_wait(self, slot):
        self._check(self._lib.wk_words_wait(self._h, int(slot)))

    def words_flush(self):
        self._check(self._lib.wk_words_flush(self._h))

    def words_pending(self):
        a, b = C.c_int64(0), C.c_int64(0)
        self._check(self._lib.wk_words_pending(self._h, C.byref(a),
                                               C.byref(b)))
        return a.value, b.value

    # -- SAM tokenizer on the device -----------------------------------------
    def dtok_copy(self, buf, begin, stop):
        """Start copying ``buf[begin:stop]`` (pinned) to the device; the
        ``dtok_scan`` of the same block finds it there."""
        raw = np.frombuffer(memoryview(buf), dtype=np.uint8)
        if raw.size:
            self._check(self._lib.wk_dtok_copy(
                self._h, C.c_void_p(raw.ctypes.data), int(begin), int(stop)))

    def dtok_copy_ahead(self, buf, begin, stop):
        """``dtok_copy`` for a reader that reuses ``buf`` before the block is
        scanned: returns the copy's ticket; after ``dtok_copy_wait(ticket)``
        the bytes may be overwritten."""
        raw = np.frombuffer(memoryview(buf), dtype=np.uint8)
        ticket = C.c_int32(-1)
        if raw.size:
            args = (self._h, C.c_void_p(raw.ctypes.data), int(begin),
                    int(stop), C.byref(ticket))
            self._check(self._lib.wk_dtok_copy_ahead(*args))
        return ticket.value

    def dtok_copy_wait(self, ticket):
        self._check(self._lib.wk_dtok_copy_wait(self._h, int(ticket)))

    def dtok_copy_drop(self):
        """Forget the blocks copied ahead that no scan has asked for."""
        self._check(self._lib.wk_dtok_copy_drop(self._h))

    def dtok_subject_map(self, table):
        """``table[id of a name the tokenizer met]`` = index of its subject
        (`--trim-sub`), -4 for a name of the `--exclude` set; None: the ids
        are the indices."""
        if table is None:
            self._check(self._lib.wk_dtok_subject_map(self._h, None, 0))
            return
        table = np.ascontiguousarray(table, dtype=np.int32)
        if table.size == 0:     # (a map that no name has entered yet)
            one = np.zeros(1, dtype=np.int32)
            self._check(self._lib.wk_dtok_subject_map(
                self._h, _ptr(one, C.c_int32), 0))
            return
        self._check(self._lib.wk_dtok_subject_map(
            self._h, _ptr(table, C.c_int32), table.size))

    def dtok_ahead_room(self):
        """Blocks a reader may copy ahead of the scans: half of the device's
        free memory in text buffers."""
        n = C.c_int32(0)
        self._check(self._lib.wk_dtok_ahead_room(self._h, C.byref(n)))
        return n.value

    def dtok_expect(self, text_bytes):
        """The blocks scanned from now on are ``text_bytes`` bytes of one
        sample in all (0: unknown): its record buffers are sized once."""
        self._check(self._lib.wk_dtok_expect(self._h, int(text_bytes)))

    def dtok_text_back(self, n):
        """The text of the block scanned last (``n`` = its length) as the
        device holds it."""
        out = np.empty(int(n), dtype=np.uint8)
        if out.size:
            self._check(self._lib.wk_dtok_text_back(
                self._h, C.c_void_p(out.ctypes.data), int(n)))
        return out

    def dtok_stage_hits(self, genome_of_subject, th):
        """The scanned block's hits ("ex" flavour) as the staged coord-match
        chunk.  Returns (status, reads, hits)."""
        g = _arr(genome_of_subject, np.int32)
        a, b, st = C.c_int64(0), C.c_int64(0), C.c_int(1)
        self._check(self._lib.wk_dtok_stage_hits(
            self._h, _ptr(g, C.c_int32), g.size, float(th), C.byref(a),
            C.byref(b), C.byref(st)))
        return st.value, a.value, b.value

    def dtok_stage_hits_append(self, genome_of_subject, th, jobs):
        """``dtok_stage_hits`` behind the hits of the blocks staged this way
        since the last count (``wk_dtok_stage_hits_append``).  Returns
        (status, reads, hits, may_wait): ``may_wait`` -- the next block may be
        staged before ``ordinal_count``."""
        g = _arr(genome_of_subject, np.int32)
        a, b, st, w = C.c_int64(0), C.c_int64(0), C.c_int(1), C.c_int(0)
        self._check(self._lib.wk_dtok_stage_hits_append(
            self._h, _ptr(g, C.c_int32), g.size, float(th), self._jobs(jobs),
            len(jobs), C.byref(a), C.byref(b), C.byref(st), C.byref(w)))
        return st.value, a.value, b.value, bool(w.value)

    def dtok_scan(self, tok, buf, begin, stop, extra=False):
        """Copy and parse ``buf[begin:stop]`` (whole lines ending at a run
        boundary: ``Tokenizer.sam_span``) on the device.  Returns (status,
        n_lines): status 0 = parsed (new subjects are in ``tok``), 1 = the
        host tokenizer takes this block."""
        raw = np.frombuffer(memoryview(buf), dtype=np.uint8)
        n, st = C.c_int64(0), C.c_int(1)
        addr = C.c_void_p(raw.ctypes.data) if raw.size \
            else C.cast(C.c_char_p(b''), C.c_void_p)
        self._check(self._lib.wk_dtok_scan(self._h, tok._h, addr, int(begin),
                                           int(stop), int(bool(extra)),
                                           C.byref(n), C.byref(st)))
        return st.value, n.value

    def dtok_scan_emit(self, tok, buf, begin, stop):
        """``dtok_scan`` + ``dtok_emit`` of the plain flavour with one wait
        (``wk_dtok_scan_emit``).  Returns (status, n_lines, n_reads or None):
        ``n_reads`` is a number when the block's records have been appended,
        ``None`` when only the scan was done."""
        raw = np.frombuffer(memoryview(buf), dtype=np.uint8)
        n, st, em = C.c_int64(0), C.c_int(1), C.c_int(0)
        a, b = C.c_int64(0), C.c_int64(0)
        addr = C.c_void_p(raw.ctypes.data) if raw.size \
            else C.cast(C.c_char_p(b''), C.c_void_p)
        self._check(self._lib.wk_dtok_scan_emit(
            self._h, tok._h, addr, int(begin), int(stop), C.byref(n),
            C.byref(st), C.byref(em), C.byref(a), C.byref(b)))
        return st.value, n.value, (a.value if em.value else None)

    def dtok_scan_emit_begin(self, tok, buf, begin, stop):
        """Launch a block's one-kernel tokenizer without waiting for it
        (``wk_dtok_scan_emit_begin``).  False: not a block for this way right
        now, nothing has happened."""
        raw = np.frombuffer(memoryview(buf), dtype=np.uint8)
        if not raw.size:
            return False
        started = C.c_int(0)
        self._check(self._lib.wk_dtok_scan_emit_begin(
            self._h, tok._h, C.c_void_p(raw.ctypes.data), int(begin),
            int(stop), C.byref(started)))
        return bool(started.value)

    def dtok_scan_emit_end(self):
        """The verdict of the oldest block under way
        (``wk_dtok_scan_emit_end``): (n_lines, n_reads) of an appended block,
        ``None`` when the kernel handed it -- and the block behind it --
        back."""
        n, st = C.c_int64(0), C.c_int(1)
        a, b = C.c_int64(0), C.c_int64(0)
        self._check(self._lib.wk_dtok_scan_emit_end(
            self._h, C.byref(n), C.byref(st), C.byref(a), C.byref(b)))
        return (n.value, a.value) if st.value == 0 else None

    def text_upload(self, buf, begin, stop):
        """(measurement) ``buf[begin:stop]`` -- a block as the host cuts it --
        to the device, to stay: a later scan of the same bytes copies
        nothing (``wk_text_upload``)."""
        raw = np.frombuffer(memoryview(buf), dtype=np.uint8)
        self._check(self._lib.wk_text_upload(
            self._h, C.c_void_p(raw.ctypes.data), int(begin), int(stop)))

    def text_clear(self):
        self._check(self._lib.wk_text_clear(self._h))

    def ordinal_chunk_counts(self):
        """(measurement) chunks ``ordinal_count`` matched sorted by genome
        stripe, chunks it matched with the gather kernels alone."""
        a, b = C.c_int64(0), C.c_int64(0)
        self._check(self._lib.wk_ordinal_chunk_counts(self._h, C.byref(a),
                                                      C.byref(b)))
        return a.value, b.value

    def dtok_fused_counts(self):
        """(measurement) blocks the one-kernel tokenizer did, blocks it handed
        back to the six kernels."""
        a, b = C.c_int64(0), C.c_int64(0)
        self._check(self._lib.wk_dtok_fused_counts(self._h, C.byref(a),
                                                   C.byref(b)))
        return a.value, b.value

    def h2d_rate(self, nbytes=64 << 20, reps=32):
        """(measurement) bytes/s of pinned host -> device copies of
        ``nbytes`` each on this box (``wk_h2d_rate``)."""
        out = C.c_double(0.0)
        self._check(self._lib.wk_h2d_rate(self._h, int(nbytes), int(reps),
                                          C.byref(out)))
        return out.value

    def dtok_format(self, fmt):
        """Format of the blocks ``dtok_scan`` is given from now on: 'sam',
        'b6o', 'paf' or 'map' (the last in the plain flavour only)."""
        self._check(self._lib.wk_dtok_format(self._h, Tokenizer.FORMATS[fmt]))

    def dtok_emit(self):
        """Group the scanned block's lines into reads and append their
        records to the accumulated packed records.  Returns (status, reads,
        records); status 1 = nothing appended (a read of more than 16
        subjects): the host tokenizer takes the block."""
        a, b, st = C.c_int64(0), C.c_int64(0), C.c_int(1)
        self._check(self._lib.wk_dtok_emit(self._h, C.byref(a), C.byref(b),
                                           C.byref(st)))
        return st.value, a.value, b.value

    def dtok_keep_reads(self, on):
        """``wk_dtok_emit`` keeps the per-read state of its block for
        ``dtok_readmap`` (read maps formatted on the device)."""
        self._check(self._lib.wk_dtok_keep_reads(self._h, int(bool(on))))

    def readmap_tables(self, job, slot_of_subject, slot_order, shown):
        """Job ``job``'s read-map tables: the taxon slot of every subject,
        the slots' order by id string, the text shown per slot (list of
        bytes)."""
        slot_of_subject = _arr(slot_of_subject, np.int32)
        slot_order = _arr(slot_order, np.int32)
        off = np.zeros(len(shown) + 1, dtype=np.uint32)
        np.cumsum([len(x) for x in shown], out=off[1:])
        self._check(self._lib.wk_readmap_tables(
            self._h, int(job), _ptr(slot_of_subject, C.c_int32),
            slot_of_subject.size, _ptr(slot_order, C.c_int32),
            _ptr(off, C.c_uint32), b''.join(shown), len(shown)))

    def dtok_readmap(self, job, out=None):
        """Read-map text of the block emitted last at job ``job`` as a
        uint8 array (``out``: a buffer to fetch it into when it is large
        enough) and whether it lies in ``out``."""
        n = C.c_int64(0)
        self._check(self._lib.wk_dtok_readmap(self._h, int(job), C.byref(n)))
        inside = out is not None and out.size >= n.value
        if not inside:
            out = np.empty(n.value, dtype=np.uint8)
        if n.value:
            self._check(self._lib.wk_dtok_readmap_fetch(
                self._h, C.c_void_p(out.ctypes.data), out.size))
        return out[:n.value], inside

    def strata_load(self, text):
        """A sample's read -> stratum map (uint8 array of its text) as the
        device's join table.  Returns None when the kernels leave the map to
        the host's join, else (labels as bytes in order of first appearance,
        their slots)."""
        text = np.ascontiguousarray(text, dtype=np.uint8)
        n_pairs, n_lab, st = C.c_int64(0), C.c_int32(0), C.c_int(1)
        self._check(self._lib.wk_strata_load(
            self._h, C.c_void_p(text.ctypes.data if text.size else 0),
            text.size, C.byref(n_pairs), C.byref(n_lab), C.byref(st)))
        if st.value != 0:
            return None
        cap = n_lab.value
        slot = np.empty(cap, dtype=np.int32)
        off = np.empty(cap, dtype=np.int64)
        ln = np.empty(cap, dtype=np.int32)
        n = C.c_int32(0)
        self._check(self._lib.wk_strata_labels(
            self._h, _ptr(slot, C.c_int32), _ptr(off, C.c_int64),
            _ptr(ln, C.c_int32), cap, C.byref(n)))
        labels = [text[o:o + k].tobytes()
                  for o, k in zip(off[:n.value].tolist(), ln[:n.value].tolist())]
        return labels, slot[:n.value]

    def strata_groups(self, slots, groups):
        slots = _arr(slots, np.int32)
        groups = _arr(groups, np.int32)
        self._check(self._lib.wk_strata_groups(
            self._h, _ptr(slots, C.c_int32), _ptr(groups, C.c_int32),
            slots.size))

    def strata_clear(self):
        self._check(self._lib.wk_strata_clear(self._h))

    def ordinal_hit_offsets(self, n_hits):
        """Offsets of every hit's genes in the staged gene lists
        (``chunk_download``): int32[n_hits + 1]."""
        out = np.empty(n_hits + 1, dtype=np.int32)
        self._check(self._lib.wk_ordinal_hit_offsets(
            self._h, _ptr(out, C.c_int32), out.size))
        return out

    def ordinal_pair_genes(self, n_pairs):
        """Gene table index of every (hit, gene) match of the staged gene
        lists (option ``gene_index_pairs``): int32[n_pairs]."""
        out = np.empty(n_pairs, dtype=np.int32)
        self._check(self._lib.wk_ordinal_pair_genes(
            self._h, _ptr(out, C.c_int32), out.size))
        return out

    def set_uniform_group(self, group):
        self._check(self._lib.wk_set_uniform_group(self._h, int(group)))

    def chunk_download(self):
        """Return (subj int32[], qoff int32[]) of the staged classify chunk."""
        nrec, nrd = C.c_int64(0), C.c_int64(0)
        self._check(self._lib.wk_chunk_download(
            self._h, None, 0, None, 0, C.byref(nrec), C.byref(nrd)))
        subj = np.empty(nrec.value, dtype=np.int32)
        qoff = np.empty(nrd.value + 1, dtype=np.int32)
        self._check(self._lib.wk_chunk_download(
            self._h, _ptr(subj, C.c_int32), subj.size, _ptr(qoff, C.c_int32),
            qoff.size, C.byref(nrec), C.byref(nrd)))
        return subj, qoff

    # -- stats / timing ---------------------------------------------------
    def stats(self):
        s = Stats()
        self._check(self._lib.wk_get_stats(self._h, C.byref(s)))
        return {'n_reads': s.n_reads, 'n_records': s.n_records,
                'n_pairs': s.n_pairs, 'table_used': s.table_used}

    def reset_stats(self):
        self._check(self._lib.wk_reset_stats(self._h))

    def timer_begin(self):
        self._check(self._lib.wk_timer_begin(self._h))

    def timer_end(self):
        self._check(self._lib.wk_timer_end(self._h))

    def timer_ms(self):
        ms = C.c_double(0)
        self._check(self._lib.wk_timer_ms(self._h, C.byref(ms)))
        return ms.value

    def profile_kernels(self, enable=True):
        self._check(self._lib.wk_profile_kernels(self._h, int(enable)))

    def last_kernel_ms(self, family):
        ms = C.c_double(0)
        self._check(self._lib.wk_last_kernel_ms(self._h, family.encode(),
                                                C.byref(ms)))
        return ms.value


WEIGHT_L = 720720       # WK_WEIGHT_L: k = 0 keys hold multiples of 1 / L
WEIGHT_MAX_K = 16
KEY_K_MASK = np.uint64(0xFFF << 49)
KEY_GROUP_SHIFT = 28        # key >> 28 = (job, k, group)


def preorder(par, root=-1):
    """(pre, size, depth) of the tree `par` (int64 parent array) through the
    native helper; raises ValueError for no / several roots and LookupError(node)
    for a node that cannot reach the root."""
    lib = load_library()
    par = np.ascontiguousarray(par, dtype=np.int64)
    n = par.size
    pre, size, depth = (np.empty(n, np.int64), np.empty(n, np.int64),
                        np.empty(n, np.int64))
    bad = C.c_int64(-1)
    rc = lib.wk_preorder(_ptr(par, C.c_int64), n, int(root),
                         _ptr(pre, C.c_int64), _ptr(size, C.c_int64),
                         _ptr(depth, C.c_int64), C.byref(bad))
    if rc == E_STATE:
        raise LookupError(int(bad.value))
    if rc != OK:
        raise ValueError('Hierarchy must have exactly one root.')
    return pre, size, depth


def build_id():
    """Digest of the sources the loaded library was compiled from."""
    return load_library().wk_build_id().decode()


def device_count():
    """HIP devices visible to this process."""
    return int(load_library().wk_device_count())


def canonical_counts(keys, vals):
    """One canonical form of a count table: every 1/k contribution with
    k <= WEIGHT_MAX_K moved under k = 0 in units of 1 / WEIGHT_L (what the
    hash-cache paths of the device produce directly), equal keys merged.
    Returns (sorted keys, values) as uint64 arrays."""
    keys = np.asarray(keys, dtype=np.uint64)
    vals = np.asarray(vals).astype(np.uint64)
    k = (keys >> np.uint64(49)) & np.uint64(MAX_K)
    small = (k >= 1) & (k <= WEIGHT_MAX_K)
    factor = np.where(small, np.uint64(WEIGHT_L) // np.maximum(k, np.uint64(1)),
                      np.uint64(1))
    merged = np.where(small, keys & ~(np.uint64(MAX_K) << np.uint64(49)), keys)
    uniq, inv = np.unique(merged, return_inverse=True)
    out = np.zeros(uniq.size, dtype=np.uint64)
    np.add.at(out, inv, vals * factor)
    return uniq, out


def counts_to_fractions(keys, vals):
    """Fold (job, k, group, feature) -> n into {(job, group, feature):
    Fraction} = sum_k n_k / k  (exact; woltka/classify.py:167-170); k = 0
    marks values in units of 1 / WEIGHT_L."""
    job, k, grp, feat = decode_keys(keys)
    res = {}
    for j, kk, g, f, n in zip(job.tolist(), k.tolist(), grp.tolist(),
                              feat.tolist(), np.asarray(vals).tolist()):
        key = (j, g, f)
        res[key] = res.get(key, 0) + Fraction(n, kk or WEIGHT_L)
    return res


class Tokenizer:
    """Native multi-threaded SAM tokenizer (host side; needs no GPU)."""

    MATE_SUFFIX = ('', '/1', '/2')

    def __init__(self, n_threads=0, exclude=None):
        self._lib = load_library()
        h = C.c_void_p()
        if self._lib.wk_tok_create(int(n_threads), C.byref(h)) != OK:
            raise RuntimeError('wk_tok_create failed')
        self._h = h
        if exclude:
            self.set_exclude(exclude)

    def set_exclude(self, exclude):
        """Subjects whose queries are dropped (an empty set: none)."""
        names = [x.encode() for x in sorted(exclude or ())]
        off = np.zeros(len(names) + 1, dtype=np.int32)
        np.cumsum([len(x) for x in names], out=off[1:])
        blob = b''.join(names)
        self._check(self._lib.wk_tok_set_exclude(
            self._h, blob, _ptr(off, C.c_int32), len(names)))

    def sam_tail(self):
        """After the final block of a SAM file parsed with ``extra`` and an
        exclusion set: the text of the reads the reference's parser yields
        once more when the file's last query was dropped (align.py:542-547);
        b'' otherwise."""
        n = C.c_int64(0)
        self._check(self._lib.wk_tok_sam_tail(self._h, None, 0, C.byref(n)))
        if n.value == 0:
            return b''
        buf = C.create_string_buffer(n.value)
        self._check(self._lib.wk_tok_sam_tail(self._h, buf, n.value,
                                              C.byref(n)))
        return buf.raw[:n.value]

    def _check(self, rc):
        if rc == OK:
            return
        msg = self._lib.wk_tok_last_error(self._h).decode()
        if rc == E_RANGE and 'mate bits' in msg:
            raise IndexError(msg)       # align.py:339 indexes a 3-tuple with 3
        raise ValueError(msg)

    def close(self):
        if getattr(self, '_h', None):
            self._lib.wk_tok_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def strata_swap(self):
        """The table `load_strata(..., ahead=True)` filled becomes the one the
        parser joins against."""
        self._check(self._lib.wk_tok_strata_swap(self._h))

    def load_strata(self, stream, block_bytes=1 << 27, ahead=False):
        """Load a read-to-stratum map (binary stream) for the sample that is
        about to be parsed; returns the label names (index = stratum id).
        ``ahead``: into the second table (while the first is in use, from
        another thread); `strata_swap` then puts it in place."""
        self._check(self._lib.wk_tok_strata_select(self._h, int(bool(ahead))))
        try:
            return self._load_strata(stream, block_bytes)
        finally:
            self._check(self._lib.wk_tok_strata_select(self._h, 0))

    def _load_strata(self, stream, block_bytes):
        self._check(self._lib.wk_tok_strata_clear(self._h))
        n_ent, n_lab = C.c_int64(0), C.c_int32(0)
        carry = b''
        while True:
            data = stream.read(block_bytes)
            buf = carry + data
            if not data:
                cut = len(buf)
            else:
                cut = buf.rfind(b'\n') + 1
            if cut:
                raw = np.frombuffer(buf, dtype=np.uint8, count=cut)
                self._check(self._lib.wk_tok_strata_load(
                    self._h, C.c_void_p(raw.ctypes.data), cut,
                    C.byref(n_ent), C.byref(n_lab)))
            carry = buf[cut:]
            if not data:
                break
        if n_ent.value == 0:
            return []
        off = np.empty(n_lab.value + 1, dtype=np.int64)
        self._check(self._lib.wk_tok_strata_labels(self._h, None,
                                                   _ptr(off, C.c_int64)))
        blob = C.create_string_buffer(max(1, int(off[-1])))
        self._check(self._lib.wk_tok_strata_labels(self._h, blob,
                                                   _ptr(off, C.c_int64)))
        raw, o = blob.raw, off.tolist()
        return [raw[o[i]:o[i + 1]].decode() for i in range(n_lab.value)]

    def new_samples(self):
        """Sample names first seen since the last call (index order)."""
        n = C.c_int32(0)
        self._check(self._lib.wk_tok_new_samples(self._h, None, None,
                                                 C.byref(n)))
        if n.value == 0:
            return []
        off = np.empty(n.value + 1, dtype=np.int64)
        self._check(self._lib.wk_tok_new_samples(
            self._h, None, _ptr(off, C.c_int64), C.byref(n)))
        blob = C.create_string_buffer(max(1, int(off[-1])))
        self._check(self._lib.wk_tok_new_samples(
            self._h, blob, _ptr(off, C.c_int64), C.byref(n)))
        raw, o = blob.raw, off.tolist()
        return [raw[o[i]:o[i + 1]].decode() for i in range(n.value)]

    FORMATS = {'sam': 0, 'map': 1, 'b6o': 2, 'paf': 3}

    @staticmethod
    def boundary(buf, pos, fmt='sam', extra=False):
        """Offset of the first line at or after ``pos`` of ``buf`` (bytes-like,
        e.g. a memory-mapped file) that starts a new run of equal query ids."""
        mv = memoryview(buf)
        n = mv.nbytes
        if pos <= 0 or n == 0:
            return 0
        if pos >= n:
            return n
        out = C.c_int64(0)
        addr = C.c_void_p(np.frombuffer(mv, dtype=np.uint8).ctypes.data)
        rc = load_library().wk_tok_boundary(Tokenizer.FORMATS[fmt], int(extra),
                                            addr, n, int(pos), C.byref(out))
        if rc != OK:
            raise ValueError('wk_tok_boundary failed')
        return out.value

    @staticmethod
    def sam_span(buf, final, in_header, fmt='sam', extra=False):
        """(ok, begin, stop, in_header_after) of a block of alignment text (SAM
        unless ``fmt`` says otherwise): the part that can be tokenised now
        (``wk_tok_span``).  ``extra``: for the "ex" parsers, to which a line
        is a row only if it has all their fields -- the last run of rows, where
        the block is cut, is theirs then (a block cut by the plain rule and
        handed to the "ex" tokenizer would lose the read in front of a line
        that only the plain parser takes for a row)."""
        raw = np.frombuffer(memoryview(buf), dtype=np.uint8)
        b, s, h = C.c_int64(0), C.c_int64(0), C.c_int(0)
        addr = C.c_void_p(raw.ctypes.data) if raw.size \
            else C.cast(C.c_char_p(b''), C.c_void_p)
        rc = load_library().wk_tok_span(
            Tokenizer.FORMATS[fmt], int(bool(extra)), addr, raw.size,
            int(bool(final)), int(bool(in_header)), C.byref(b), C.byref(s),
            C.byref(h))
        return rc == OK, b.value, s.value, bool(h.value)

    def set_header_state(self, in_header):
        self._check(self._lib.wk_tok_set_header_state(self._h,
                                                      int(bool(in_header))))

    def read_into(self, fd, offset, view):
        """Fill ``view`` (writable bytes-like) from file descriptor ``fd`` at
        ``offset`` with all tokenizer threads; returns the bytes read (short
        only at the end of the file)."""
        mv = memoryview(view)
        n = mv.nbytes
        if n == 0:
            return 0
        raw = np.frombuffer(mv, dtype=np.uint8)
        got = C.c_int64(0)
        try:
            self._check(self._lib.wk_tok_read(
                self._h, int(fd), int(offset), C.c_void_p(raw.ctypes.data), n,
                C.byref(got)))
        finally:
            del raw
            mv.release()
        return got.value

    def trim(self, src, begin, want, keep_tabs, dst, size=None):
        """SAM text ``[begin, begin + want)`` (``begin`` a line start) of
        ``src`` -- a uint8 array (the file, mapped) or an open file descriptor
        with its ``size`` -- cut to its first ``keep_tabs`` columns into
        ``dst`` by all tokenizer threads (``wk_tok_trim``).  Returns (input
        bytes taken, bytes written): whole lines only."""
        raw = np.frombuffer(memoryview(dst), dtype=np.uint8)
        consumed, got = C.c_int64(0), C.c_int64(0)
        if isinstance(src, int):
            mem, fd, n = None, src, int(size)
        else:
            mem, fd, n = C.c_void_p(src.ctypes.data), -1, int(src.size)
        try:
            self._check(self._lib.wk_tok_trim(
                self._h, mem, fd, n, int(begin), int(want), int(keep_tabs),
                C.c_void_p(raw.ctypes.data), int(raw.size), C.byref(consumed),
                C.byref(got)))
        finally:
            del raw
        return consumed.value, got.value

    def set_subject_map(self, table):
        """Translate subject ids at fetch time: ``subj`` comes out as
        ``table[id]`` (-1 beyond the table); ``None`` switches it off."""
        if table is None:
            self._check(self._lib.wk_tok_set_subject_map(self._h, None, 0))
            return
        table = _arr(table, np.int32)
        self._check(self._lib.wk_tok_set_subject_map(
            self._h, _ptr(table, C.c_int32), table.size))

    def parse(self, buf, first=False, final=False, extra=False,
              want_names=False, want_groups=False, want_samples=False,
              fmt='sam', sink=None):
        """Tokenize ``buf`` (bytes-like) of alignment format ``fmt``
        (sam / map / b6o / paf).  Returns a dict with ``consumed``,
        ``subj``, ``off`` (+ ``beg``/``end``/``len`` with ``extra``,
        ``qname`` descriptors with ``want_names``, ``group`` = stratum ids with
        ``want_groups``).

        ``sink(tok, n_reads, n_records)`` is called between tokenising and
        fetching and may return the arrays the results are written into
        (pinned staging buffers): ``{'packed': uint32[]}`` asks for the plain
        flavour's records as packed words (``result['words']``; blocks with a
        read of more than 16 records come back the general way), ``{'subj',
        'off', 'beg', 'end', 'len'}`` for the arrays themselves; anything it
        returns is marked ``result['sunk']``."""
        mv = memoryview(buf)
        n = mv.nbytes
        addr = C.c_void_p(np.frombuffer(mv, dtype=np.uint8).ctypes.data) \
            if n else C.c_void_p(0)
        if n == 0:
            addr = C.cast(C.c_char_p(b''), C.c_void_p)
        consumed, nrd, nrec = C.c_int64(), C.c_int64(), C.c_int64()
        if fmt == 'map':
            extra = False           # no "ex" flavour (align.py:236)
        self._check(self._lib.wk_tok_text(
            self._h, self.FORMATS[fmt], addr, n, int(first), int(final),
            int(extra),
            int(bool(want_names)) | (2 if want_groups else 0) |
            (4 if want_samples else 0),
            C.byref(consumed), C.byref(nrd), C.byref(nrec)))
        bufs = sink(self, nrd.value, nrec.value) if sink is not None else None
        packed_out = bufs.get('packed') if bufs else None
        if packed_out is not None and not extra and not want_names and \
                not want_groups and not want_samples and \
                nrec.value <= packed_out.size:
            # the records as packed words, written by all tokenizer threads
            # straight into the caller's (pinned) buffer; blocks with reads
            # of more than 16 records take the general route below
            n_big = C.c_int64(0)
            rc = self._lib.wk_tok_fetch_packed(
                self._h, _ptr(packed_out, C.c_uint32), None, None,
                C.byref(n_big))
            if rc == OK and n_big.value == 0:
                return {'consumed': consumed.value, 'sunk': True,
                        'words': packed_out[:nrec.value],
                        'n_reads': nrd.value}
            if rc not in (OK, E_RANGE):
                self._check(rc)
        out = {'consumed': consumed.value}
        sunk = bool(bufs) and 'subj' in bufs and \
            bufs['subj'].size >= nrec.value and \
            bufs['off'].size >= nrd.value + 1 and \
            (not extra or min(bufs[k].size for k in ('beg', 'end', 'len'))
             >= nrec.value)
        if sunk:
            out['sunk'] = True
            out['subj'] = bufs['subj'][:nrec.value]
            out['off'] = bufs['off'][:nrd.value + 1]
            if extra:
                for k in ('beg', 'end', 'len'):
                    out[k] = bufs[k][:nrec.value]
        else:
            out['subj'] = np.empty(nrec.value, np.int32)
            out['off'] = np.empty(nrd.value + 1, np.int32)
            if extra:
                out['beg'] = np.empty(nrec.value, np.int32)
                out['end'] = np.empty(nrec.value, np.int32)
                out['len'] = np.empty(nrec.value, np.uint32)
        if want_names:
            out['qname'] = np.empty(nrd.value, np.uint64)
        self._check(self._lib.wk_tok_fetch(
            self._h, _ptr(out['subj'], C.c_int32), _ptr(out['off'], C.c_int32),
            _ptr(out.get('beg'), C.c_int32), _ptr(out.get('end'), C.c_int32),
            _ptr(out.get('len'), C.c_uint32),
            _ptr(out.get('qname'), C.c_uint64)))
        if want_samples:
            out['sample'] = np.empty(nrd.value, np.int32)
            self._check(self._lib.wk_tok_fetch_samples(
                self._h, _ptr(out['sample'], C.c_int32)))
        if want_groups:
            out['group'] = np.empty(nrd.value, np.int32)
            self._check(self._lib.wk_tok_fetch_groups(
                self._h, _ptr(out['group'], C.c_int32)))
        return out

    def new_subjects(self):
        """Names of the subjects first seen since the last call, in index
        order."""
        tot, new, nbytes = C.c_int32(), C.c_int32(), C.c_int64()
        self._check(self._lib.wk_tok_subjects(
            self._h, C.byref(tot), C.byref(new), C.byref(nbytes)))
        if new.value == 0:
            return []
        blob = C.create_string_buffer(max(1, nbytes.value))
        off = np.empty(new.value + 1, dtype=np.int32)
        self._check(self._lib.wk_tok_new_subjects(self._h, blob,
                                                  _ptr(off, C.c_int32)))
        raw = blob.raw
        o = off.tolist()
        return [raw[o[i]:o[i + 1]].decode() for i in range(new.value)]

    @classmethod
    def query_names(cls, buf, desc):
        """Read ids from QNAME descriptors (QNAME + '', '/1' or '/2')."""
        mv = buf if isinstance(buf, bytes) else bytes(buf)
        out = []
        for d in desc.tolist():
            o, ln, m = d >> 24, (d >> 2) & 0x3FFFFF, d & 3
            out.append(mv[o:o + ln].decode() + cls.MATE_SUFFIX[m])
        return out


def gz_member(data):
    """One gzip member (bytes) holding ``data`` (any buffer), deflated by
    ``wk_gz_member`` (csrc/wk_deflate.cpp); the GIL is released meanwhile."""
    lib = load_library()
    raw = np.frombuffer(memoryview(data), dtype=np.uint8)
    cap = lib.wk_gz_bound(raw.size)
    out = np.empty(cap, dtype=np.uint8)
    n = lib.wk_gz_member(C.c_void_p(raw.ctypes.data if raw.size else 0),
                         raw.size, C.c_void_p(out.ctypes.data), cap)
    if n < 0:
        raise ValueError('wk_gz_member failed')
    return out[:n].tobytes()


class Gunzip:
    """A regular gzip file inflated by ``wk_gunzip_*`` (csrc/wk_inflate.cpp):
    ``readinto(buffer)`` fills a writable buffer of at least 64 KB with the
    next bytes of text on ``threads`` threads (the GIL is released) and
    returns their number, 0 at the end.  ``OSError`` where the gzip module
    raises one (damaged data, CRC); ``ValueError`` when the file is not what
    this reader takes (the caller then opens it the ordinary way)."""

    def __init__(self, path, threads=1):
        self._lib = load_library()
        err = C.create_string_buffer(200)
        self._h = self._lib.wk_gunzip_open(os.fsencode(path), int(threads),
                                           err, len(err))
        if not self._h:
            raise ValueError(err.value.decode() or 'wk_gunzip_open failed')

    def readinto(self, out):
        arr = np.frombuffer(out, dtype=np.uint8) \
            if not isinstance(out, np.ndarray) else out
        n = self._lib.wk_gunzip_read(self._h, C.c_void_p(arr.ctypes.data),
                                     arr.size)
        if n < 0:
            raise OSError(self._lib.wk_gunzip_error(self._h).decode())
        return n

    def close(self):
        if getattr(self, '_h', None):
            self._lib.wk_gunzip_close(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def gz_inflate_members(blob, spans, out=None, n_threads=0):
    """The text of a chain of 'WK' gzip members (``pgzip.members_of``) as one
    uint8 array, inflated on ``n_threads`` threads; ``out``: a buffer to
    inflate into when it is large enough.  Returns (text, inside out?)."""
    lib = load_library()
    raw = np.frombuffer(memoryview(blob), dtype=np.uint8)
    lo = np.asarray([a for a, _ in spans], dtype=np.int64)
    hi = np.asarray([b for _, b in spans], dtype=np.int64)
    isize = np.empty(lo.size, dtype=np.int64)
    for i, b in enumerate(hi.tolist()):
        isize[i] = int.from_bytes(raw[b - 4:b].tobytes(), 'little')
    off = np.zeros(lo.size + 1, dtype=np.int64)
    np.cumsum(isize, out=off[1:])
    total = int(off[-1])
    inside = out is not None and out.size >= total
    if not inside:
        out = np.empty(total, dtype=np.uint8)
    rc = lib.wk_gz_inflate_members(
        C.c_void_p(raw.ctypes.data if raw.size else 0), _ptr(lo, C.c_int64),
        _ptr(hi, C.c_int64), lo.size, C.c_void_p(out.ctypes.data),
        _ptr(off, C.c_int64), int(n_threads) or (os.cpu_count() or 1))
    if rc != 0:
        raise OSError(f'gzip member {-rc - 1} does not inflate to what its '
                      'trailer says')
    return out[:total], inside


def crc32(data, crc=0):
    lib = load_library()
    raw = np.frombuffer(memoryview(data), dtype=np.uint8)
    return lib.wk_crc32(crc, C.c_void_p(raw.ctypes.data if raw.size else 0),
                        raw.size)


def format_readmap(buf, qname, assign, m_off, m_feat, m_count, names,
                   unassigned=False, n_threads=0):
    """Read-map text (bytes) of one chunk through ``wk_format_readmap``.
    ``names[f]`` is the text printed for feature id ``f`` (list of str)."""
    lib = load_library()
    enc = [x.encode() for x in names]
    noff = np.zeros(len(enc) + 1, dtype=np.int64)
    np.cumsum([len(x) for x in enc], out=noff[1:])
    blob = b''.join(enc)
    qname = _arr(qname, np.uint64)
    assign = _arr(assign, np.int32)
    m_off = _arr(m_off, np.int64)
    m_feat = _arr(m_feat, np.int32)
    m_count = _arr(m_count, np.int32)
    raw = np.frombuffer(memoryview(buf), dtype=np.uint8)
    text = C.c_void_p(raw.ctypes.data if raw.size else 0)
    if raw.size == 0:
        text = C.cast(C.c_char_p(b''), C.c_void_p)
    n = C.c_int64(0)
    args = (text, _ptr(qname, C.c_uint64), _ptr(assign, C.c_int32),
            assign.size, _ptr(m_off, C.c_int64), _ptr(m_feat, C.c_int32),
            _ptr(m_count, C.c_int32), blob, _ptr(noff, C.c_int64), len(enc),
            int(bool(unassigned)), int(n_threads))
    if lib.wk_format_readmap(*args, None, 0, C.byref(n)) != OK:
        raise ValueError('wk_format_readmap: bad arguments')
    out = np.empty(n.value, dtype=np.uint8)
    if n.value and lib.wk_format_readmap(
            *args, C.c_void_p(out.ctypes.data), n.value, C.byref(n)) != OK:
        raise RuntimeError('wk_format_readmap failed')
    return out.tobytes()


HIER_NODES, HIER_MAP, HIER_NAMES = 0, 1, 2
HIER_PARENT, HIER_RANK, HIER_NAME = 0, 1, 2


def _blob(strings):
    """(bytes, int64 offsets) of a sequence of str."""
    enc = [x.encode() for x in strings]
    off = np.zeros(len(enc) + 1, dtype=np.int64)
    if enc:
        np.cumsum(np.fromiter(map(len, enc), np.int64, len(enc)), out=off[1:])
    return b''.join(enc), off


def table_body(keys, values, n_threads=8):
    """(bytes-like body, rows) of a one-sample TSV table: ``keys`` = the feature
    ids joined by ``\\n`` (bytes), ``values`` their int64 values
    (``wk_table_body``).  None when the native side refuses the input."""
    values = _arr(values, np.int64)
    n = values.size
    out = np.empty(len(keys) + 24 * n + 8, dtype=np.uint8)
    used, rows = np.zeros(1, np.int64), np.zeros(1, np.int64)
    rc = load_library().wk_table_body(
        keys, len(keys), _ptr(values, C.c_int64), n, int(n_threads),
        C.c_void_p(out.ctypes.data), out.size, _ptr(used, C.c_int64),
        _ptr(rows, C.c_int64))
    if rc != OK:
        return None
    return memoryview(out)[:int(used[0])], int(rows[0])


def table_rows(prefixes, names, prefix_of_row, name_of_row, values):
    """(bytes-like body, rows) of a TSV table whose rows are named
    ``prefix|name`` (``prefixes``: list of str or None) with ``values`` =
    int64[n_rows, n_cols] (``wk_table_rows``: sorted, all-zero rows left out).
    None when the native side refuses the input (a name with a line break)."""
    values = np.ascontiguousarray(values, dtype=np.int64)
    n_rows, n_cols = values.shape
    try:
        pb = '\n'.join(prefixes).encode() if prefixes else b''
        nb = '\n'.join(names).encode()
    except UnicodeEncodeError:
        return None
    if (prefixes and pb.count(b'\n') != len(prefixes) - 1) or \
            (names and nb.count(b'\n') != len(names) - 1):
        return None
    name_of_row = _arr(name_of_row, np.int32)
    if prefixes:
        prefix_of_row = _arr(prefix_of_row, np.int32)
        plen = np.fromiter(map(len, prefixes), np.int64, len(prefixes))
        per_row = int(np.where(prefix_of_row >= 0,
                               plen[np.maximum(prefix_of_row, 0)], 0).sum())
    else:
        prefix_of_row, per_row = None, 0
    nlen = np.fromiter(map(len, names), np.int64, len(names))
    # (lengths in characters: a character takes up to four bytes)
    cap = 4 * (per_row + int(nlen[name_of_row].sum())) + \
        n_rows * (3 + 21 * n_cols) + 64
    out = np.empty(cap, dtype=np.uint8)
    used, rows = np.zeros(1, np.int64), np.zeros(1, np.int64)
    rc = load_library().wk_table_rows(
        pb, len(pb), len(prefixes) if prefixes else 0, nb, len(nb), len(names),
        _ptr(prefix_of_row, C.c_int32), _ptr(name_of_row, C.c_int32),
        _ptr(values, C.c_int64), n_rows, n_cols, C.c_void_p(out.ctypes.data),
        out.size, _ptr(used, C.c_int64), _ptr(rows, C.c_int64))
    if rc != OK:
        return None
    return memoryview(out)[:int(used[0])], int(rows[0])


def _split(raw, off):
    """list of str from a bytes blob and its offsets."""
    n = len(off) - 1
    if n > 1000 and b'\n' not in raw:
        # one decode + one split instead of a slice and a decode per string
        off = _arr(off, np.int64)
        out = np.empty(int(off[-1] - off[0]) + n, dtype=np.uint8)
        if load_library().wk_blob_join(raw, _ptr(off, C.c_int64), n, b'\n',
                                       C.c_void_p(out.ctypes.data)) == OK:
            return out.tobytes().decode().split('\n')[:-1]
    o = off.tolist() if hasattr(off, 'tolist') else list(off)
    return [raw[a:b].decode() for a, b in zip(o, o[1:])]


class HierarchyBuilder:
    """Native hierarchy ingest (``wk_hier_*``; host side, needs no GPU): the
    reference's three dicts — child -> parent, node -> rank, node -> name — as
    one native symbol table.  ``add_text`` / ``update`` are one
    ``util.update_dict`` each; ``finish`` is ``tree.fill_root`` + flattening."""

    class Refused(Exception):
        """The native reader leaves this text to the Python reader."""

    def __init__(self, n_threads=0):
        self._lib = load_library()
        h = C.c_void_p()
        if self._lib.wk_hier_create(int(n_threads), C.byref(h)) != OK:
            raise RuntimeError('wk_hier_create failed')
        self._h = h
        self.n_nodes = None

    def close(self):
        if getattr(self, '_h', None):
            self._lib.wk_hier_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc == OK:
            return
        msg = self._lib.wk_hier_last_error(self._h).decode()
        if rc == E_STATE and msg.startswith('Conflicting'):
            raise AssertionError(msg)       # util.update_dict (util.py:46-75)
        if rc == E_RANGE and 'index' in msg:
            raise IndexError(msg)           # x[1] of a short line (tree.py:96)
        raise ValueError(msg)

    def add_text(self, kind, buf, rank=None):
        """One file's bytes (bytes-like).  Raises ``Refused`` when the text
        has to go through the Python reader."""
        mv = memoryview(buf)
        n = mv.nbytes
        raw = np.frombuffer(mv, dtype=np.uint8) if n else None
        try:
            addr = C.c_void_p(raw.ctypes.data) if n \
                else C.cast(C.c_char_p(b''), C.c_void_p)
            rc = self._lib.wk_hier_add_text(
                self._h, kind, addr, n,
                None if rank is None else rank.encode())
        finally:            # (no export of the caller's buffer outlives the call)
            del raw
            mv.release()
        if rc == E_ARG:
            raise self.Refused()
        self._check(rc)

    def update(self, field, pairs):
        """``update_dict`` with a dict (values may be None for the tree)."""
        keys = list(pairs)
        vals = [pairs[k] for k in keys]
        none = np.fromiter((v is None for v in vals), np.uint8, len(vals))
        kblob, koff = _blob(keys)
        vblob, voff = _blob(['' if v is None else v for v in vals])
        self._check(self._lib.wk_hier_update(
            self._h, field, kblob, _ptr(koff, C.c_int64), vblob,
            _ptr(voff, C.c_int64), _ptr(none, C.c_uint8), len(keys)))

    def finish(self):
        n, nr = C.c_int64(0), C.c_int32(0)
        self._check(self._lib.wk_hier_finish(self._h, C.byref(n),
                                             C.byref(nr)))
        self.n_nodes, self.n_ranks = n.value, nr.value
        return self.n_nodes

    def arrays(self):
        """(parent, last, rank_code, depth), int32, pre-order."""
        out = [np.empty(self.n_nodes, np.int32) for _ in range(4)]
        self._check(self._lib.wk_hier_arrays(
            self._h, *(_ptr(a, C.c_int32) for a in out)))
        return out

    def ranks(self):
        """(rank names in code order — code = index + 1 —, keys per rank)."""
        off = np.zeros(self.n_ranks + 1, np.int64)
        used = np.zeros(self.n_ranks, np.int64)
        self._check(self._lib.wk_hier_ranks(self._h, None, 0,
                                            _ptr(off, C.c_int64), None))
        blob = C.create_string_buffer(max(1, int(off[-1])))
        self._check(self._lib.wk_hier_ranks(
            self._h, blob, int(off[-1]), _ptr(off, C.c_int64),
            _ptr(used, C.c_int64)))
        return _split(blob.raw, off), used.tolist()

    def lookup(self, names):
        """Pre-order ids (int32 array, -1 = not a node) of a list of str."""
        blob, off = _blob(names)
        out = np.empty(len(names), np.int32)
        self._check(self._lib.wk_hier_lookup(
            self._h, blob, _ptr(off, C.c_int64), len(names),
            _ptr(out, C.c_int32)))
        return out

    def node_names(self, ids):
        """Names (list of str) of pre-order ids."""
        ids = _arr(ids, np.int32)
        off = np.zeros(ids.size + 1, np.int64)
        self._check(self._lib.wk_hier_node_names(
            self._h, _ptr(ids, C.c_int32), ids.size, None, 0,
            _ptr(off, C.c_int64)))
        blob = C.create_string_buffer(max(1, int(off[-1])))
        self._check(self._lib.wk_hier_node_names(
            self._h, _ptr(ids, C.c_int32), ids.size, blob, int(off[-1]),
            _ptr(off, C.c_int64)))
        return _split(blob.raw, off)

    def get(self, field, key):
        """Value of one dict entry (str), or None when the key is absent."""
        k = key.encode()
        n = C.c_int64(0)
        buf = C.create_string_buffer(256)
        self._check(self._lib.wk_hier_get(self._h, field, k, len(k), buf, 256,
                                          C.byref(n)))
        if n.value < 0:
            return None
        if n.value > 256:
            buf = C.create_string_buffer(n.value)
            self._check(self._lib.wk_hier_get(self._h, field, k, len(k), buf,
                                              n.value, C.byref(n)))
        return buf.raw[:n.value].decode()

    def size(self, field):
        return int(self._lib.wk_hier_size(self._h, field))

    def keys(self, field):
        n = self.size(field)
        off = np.zeros(n + 1, np.int64)
        self._check(self._lib.wk_hier_keys(self._h, field, None, 0,
                                           _ptr(off, C.c_int64)))
        blob = C.create_string_buffer(max(1, int(off[-1])))
        self._check(self._lib.wk_hier_keys(self._h, field, blob,
                                           int(off[-1]), _ptr(off, C.c_int64)))
        return _split(blob.raw, off)


def parse_gene_coords(buf):
    """Gene coordinates text (bytes-like) through ``wk_coords_parse``.
    Returns (goff, start0, end, genome names, (gene blob, gene offsets),
    isdup), or None when the text is left to the Python reader; raises
    ValueError with the reference's message for a malformed file."""
    lib = load_library()
    mv = memoryview(buf)
    n = mv.nbytes
    raw = np.frombuffer(mv, dtype=np.uint8) if n else None
    h = C.c_void_p()
    try:
        addr = C.c_void_p(raw.ctypes.data) if n \
            else C.cast(C.c_char_p(b''), C.c_void_p)
        rc = lib.wk_coords_parse(addr, n, C.byref(h))
        if rc == E_STATE:
            return None
        if rc != OK:
            raise ValueError(lib.wk_coords_error(h).decode(errors='replace'))
        ng, nn, gb, nb, dup = (C.c_int32(), C.c_int32(), C.c_int64(),
                               C.c_int64(), C.c_int())
        lib.wk_coords_sizes(h, C.byref(ng), C.byref(nn), C.byref(gb),
                            C.byref(nb), C.byref(dup))
        goff = np.empty(ng.value + 1, np.int32)
        start0, end = np.empty(nn.value, np.int32), np.empty(nn.value, np.int32)
        findex = np.empty(nn.value, np.int32)
        gblob = C.create_string_buffer(max(1, gb.value))
        nblob = C.create_string_buffer(max(1, nb.value))
        g_off = np.empty(ng.value + 1, np.int64)
        n_off = np.empty(nn.value + 1, np.int64)
        lib.wk_coords_fetch(h, _ptr(goff, C.c_int32), _ptr(start0, C.c_int32),
                            _ptr(end, C.c_int32), _ptr(findex, C.c_int32), gblob,
                            _ptr(g_off, C.c_int64), nblob, _ptr(n_off, C.c_int64))
        return (goff, start0, end, _split(gblob.raw, g_off),
                (nblob.raw[:nb.value], n_off), bool(dup.value), findex)
    finally:
        if h:
            lib.wk_coords_free(h)
        del raw
        mv.release()

