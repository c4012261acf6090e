"""ctypes binding of include/strus_pattern_amd.h (the C-ABI of libstruspattern_amd.so).

The library is the product: if it cannot be loaded this module raises -- there is no Python or
CPU fallback for the match path."""
import ctypes
import os

from . import build as _build

c_u32 = ctypes.c_uint32
c_u64 = ctypes.c_uint64
c_vp = ctypes.c_void_p
c_cp = ctypes.c_char_p
P = ctypes.POINTER


class SpLexem(ctypes.Structure):
    _fields_ = [("id", c_u32), ("ordpos", c_u32), ("origpos", c_u32), ("origsize", c_u32)]


class SpResult(ctypes.Structure):
    _fields_ = [(n, c_u32) for n in ("handle", "ordpos", "ordend", "origseg", "origpos", "origendseg", "origend", "item_begin", "item_count")]


class SpResultItem(ctypes.Structure):
    _fields_ = [(n, c_u32) for n in ("variable", "ordpos", "ordend", "origseg", "origpos", "origendseg", "origend")]


class SpMatcherStats(ctypes.Structure):
    _fields_ = [(n, ctypes.c_double) for n in ("nofProgramsInstalled", "nofAltKeyProgramsInstalled", "nofSignalsFired", "nofTriggersAvgActive")]


class SpMatchBatch(ctypes.Structure):
    _fields_ = [
        ("ndocs", ctypes.c_size_t), ("nresults", ctypes.c_size_t), ("nitems", ctypes.c_size_t),
        ("results", P(SpResult)), ("items", P(SpResultItem)),
        ("doc_result_offsets", P(c_u64)), ("doc_stats", P(c_u64)), ("doc_status", P(ctypes.c_int32)),
        ("result_format", P(c_u32)), ("item_format", P(c_u32)),
    ]


class SpLexBatch(ctypes.Structure):
    _fields_ = [
        ("ndocs", ctypes.c_size_t), ("nlexems", ctypes.c_size_t), ("lexems", P(SpLexem)),
        ("doc_lexem_offsets", P(c_u64)), ("doc_status", P(ctypes.c_int32)),
    ]


class SpLexDeviceBatch(ctypes.Structure):
    _fields_ = [("ndocs", ctypes.c_size_t), ("d_lexems", c_vp), ("d_doc_ranges", c_vp), ("d_doc_status", c_vp), ("d_counters", c_vp)]


class SpMatchDeviceBatch(ctypes.Structure):
    _fields_ = [
        ("ndocs", ctypes.c_size_t), ("d_results", c_vp), ("d_items", c_vp), ("d_doc_result_offsets", c_vp),
        ("d_doc_stats", c_vp), ("d_doc_status", c_vp), ("d_counters", c_vp),
        ("d_result_format", c_vp), ("d_item_format", c_vp),
    ]


# name -> (restype, argtypes); every symbol declared in include/strus_pattern_amd.h
SIGNATURES = {
    "sp_version": (c_cp, []),
    "sp_device_count": (ctypes.c_int, []),
    "sp_free": (None, [c_vp]),
    "sp_test_fail_alloc_above": (None, [c_u64]),
    "sp_matcher_create": (c_vp, []),
    "sp_matcher_free": (None, [c_vp]),
    "sp_matcher_last_error": (c_cp, [c_vp]),
    "sp_matcher_define_term_frequency": (ctypes.c_int, [c_vp, c_u32, ctypes.c_double]),
    "sp_matcher_push_term": (ctypes.c_int, [c_vp, c_u32]),
    "sp_matcher_push_expression": (ctypes.c_int, [c_vp, ctypes.c_int, ctypes.c_size_t, c_u32, c_u32]),
    "sp_matcher_push_pattern": (ctypes.c_int, [c_vp, c_cp]),
    "sp_matcher_attach_variable": (ctypes.c_int, [c_vp, c_cp]),
    "sp_matcher_define_pattern": (ctypes.c_int, [c_vp, c_cp, c_cp, ctypes.c_int]),
    "sp_matcher_define_option": (ctypes.c_int, [c_vp, c_cp, ctypes.c_double]),
    "sp_matcher_compile": (ctypes.c_int, [c_vp]),
    "sp_matcher_pattern_id": (c_u32, [c_vp, c_cp]),
    "sp_matcher_pattern_name": (c_cp, [c_vp, c_u32]),
    "sp_matcher_variable_id": (c_u32, [c_vp, c_cp]),
    "sp_matcher_variable_name": (c_cp, [c_vp, c_u32]),
    "sp_matcher_dump_table": (ctypes.c_size_t, [c_vp, P(P(c_u32))]),
    "sp_matcher_fast_tier": (ctypes.c_int, [c_vp, ctypes.c_char_p, ctypes.c_size_t]),
    "sp_matcher_serialize": (ctypes.c_int, [c_vp, P(c_vp), P(ctypes.c_size_t)]),
    "sp_matcher_deserialize": (c_vp, [c_vp, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]),
    "sp_lexer_serialize": (ctypes.c_int, [c_vp, P(c_vp), P(ctypes.c_size_t)]),
    "sp_lexer_deserialize": (c_vp, [c_vp, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]),
    "sp_matcher_format_count": (c_u32, [c_vp]),
    "sp_matcher_format_string": (c_cp, [c_vp, c_u32]),
    "sp_matcher_ctx_fetch_formats": (ctypes.c_int, [c_vp, P(P(c_u32)), P(P(c_u32))]),
    "sp_matcher_ctx_create": (c_vp, [c_vp, ctypes.c_int]),
    "sp_matcher_ctx_free": (None, [c_vp]),
    "sp_matcher_ctx_last_error": (c_cp, [c_vp]),
    "sp_matcher_ctx_put_input": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_size_t]),
    "sp_matcher_ctx_fetch_results": (ctypes.c_int, [c_vp, P(P(SpResult)), P(ctypes.c_size_t), P(P(SpResultItem)), P(ctypes.c_size_t)]),
    "sp_matcher_ctx_statistics": (ctypes.c_int, [c_vp, P(SpMatcherStats)]),
    "sp_matcher_ctx_reset": (ctypes.c_int, [c_vp]),
    "sp_matcher_ctx_match_docs": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, ctypes.c_size_t, P(SpMatchBatch)]),
    "sp_match_batch_free": (None, [P(SpMatchBatch)]),
    "sp_matcher_ctx_match_docs_device": (ctypes.c_int, [c_vp, c_vp, c_vp, c_vp, ctypes.c_size_t, ctypes.c_size_t, c_vp, P(SpMatchDeviceBatch)]),
    "sp_matcher_ctx_match_lexed_device": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_size_t, ctypes.c_size_t, c_vp, P(SpMatchDeviceBatch)]),
    "sp_matcher_ctx_batch_fetch": (ctypes.c_int, [c_vp, P(SpMatchBatch)]),
    "sp_matcher_ctx_batch_fetch_docs": (ctypes.c_int, [c_vp, ctypes.c_size_t, ctypes.c_size_t, P(SpMatchBatch)]),
    "sp_matcher_ctx_batch_counters": (ctypes.c_int, [c_vp, P(c_u64)]),
    "sp_matcher_ctx_last_kernel_ms": (ctypes.c_double, [c_vp]),
    "sp_matcher_ctx_kernel_kind": (ctypes.c_int, [c_vp]),
    "sp_matcher_ctx_kernel_name": (ctypes.c_char_p, [c_vp]),
    "sp_matcher_ctx_batch_status": (ctypes.c_int, [c_vp, c_vp, ctypes.c_size_t]),
    "sp_matcher_ctx_grow_arena": (ctypes.c_int, [c_vp]),
    "sp_matcher_ctx_reserve_output": (ctypes.c_int, [c_vp, c_u64, c_u64]),
    "sp_matcher_ctx_set_arena": (ctypes.c_int, [c_vp, c_u32, c_u32, c_u32, c_u32, c_u32]),
    "sp_lexer_create": (c_vp, []),
    "sp_lexer_free": (None, [c_vp]),
    "sp_lexer_last_error": (c_cp, [c_vp]),
    "sp_lexer_define_lexem_name": (ctypes.c_int, [c_vp, c_u32, c_cp]),
    "sp_lexer_get_lexem_name": (c_cp, [c_vp, c_u32]),
    "sp_lexer_define_lexem": (ctypes.c_int, [c_vp, c_u32, c_cp, c_u32, c_u32, ctypes.c_int]),
    "sp_lexer_define_symbol": (ctypes.c_int, [c_vp, c_u32, c_u32, c_cp]),
    "sp_lexer_get_symbol": (c_u32, [c_vp, c_u32, c_cp]),
    "sp_lexer_define_option": (ctypes.c_int, [c_vp, c_cp, ctypes.c_double]),
    "sp_lexer_compile": (ctypes.c_int, [c_vp]),
    "sp_lexer_dump_tables": (ctypes.c_size_t, [c_vp, P(P(c_u64))]),
    "sp_lexer_ctx_create": (c_vp, [c_vp, ctypes.c_int]),
    "sp_lexer_ctx_free": (None, [c_vp]),
    "sp_lexer_ctx_last_error": (c_cp, [c_vp]),
    "sp_lexer_ctx_match": (ctypes.c_int, [c_vp, c_cp, ctypes.c_size_t, P(P(SpLexem)), P(ctypes.c_size_t)]),
    "sp_lexer_ctx_reset": (ctypes.c_int, [c_vp]),
    "sp_lexer_ctx_match_docs": (ctypes.c_int, [c_vp, c_cp, c_vp, ctypes.c_size_t, P(SpLexBatch)]),
    "sp_lex_batch_free": (None, [P(SpLexBatch)]),
    "sp_lexer_ctx_match_docs_device": (ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_size_t, ctypes.c_size_t, c_vp, P(SpLexDeviceBatch)]),
    "sp_lexer_ctx_batch_fetch_docs": (ctypes.c_int, [c_vp, ctypes.c_size_t, ctypes.c_size_t, P(SpLexBatch)]),
    "sp_lexer_ctx_batch_counters": (ctypes.c_int, [c_vp, P(c_u64)]),
    "sp_lexer_ctx_batch_status": (ctypes.c_int, [c_vp, c_vp, ctypes.c_size_t]),
    "sp_lexer_ctx_last_kernel_ms": (ctypes.c_double, [c_vp]),
    "sp_lexer_ctx_last_kernel_ms_split": (ctypes.c_int, [c_vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "sp_lexer_ctx_scan_kernel_name": (ctypes.c_char_p, [c_vp]),
    "sp_lexer_ctx_words_kernel_name": (ctypes.c_char_p, [c_vp]),
    "sp_lexer_ctx_last_kernel_ms_split3": (ctypes.c_int, [c_vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "sp_lexer_ctx_reserve_output": (ctypes.c_int, [c_vp, c_u64]),
    "sp_lexer_ctx_grow_arena": (ctypes.c_int, [c_vp]),
}

_LIB = None


def lib():
    """Loads libstruspattern_amd.so (building it first if the sources are newer)."""
    global _LIB
    if _LIB is None:
        path = _build.LIB
        if _build.needs_build():
            path = _build.build()
        L = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB
