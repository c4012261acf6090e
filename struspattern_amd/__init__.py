"""struspattern_amd -- MI355X-native two-level pattern engine behind the strus pattern API.

Python mirror of the reference's plugin interfaces for the hot path (same method names and
argument meaning as src/patternMatcher.cpp / src/patternLexer.cpp of strusPattern), layered on
the C-ABI of libstruspattern_amd.so (include/strus_pattern_amd.h).  The native library is the
product; there is no Python/CPU fallback for matching.
"""
import ctypes

import numpy as np

from . import capi
from .capi import SpLexem, SpResult, SpResultItem

__all__ = ["PatternMatcher", "PatternMatcherInstance", "PatternMatcherContext", "PatternLexer", "PatternLexerInstance",
           "PatternLexerContext", "PatternError", "JOIN_OP", "POSITION_BIND"]

JOIN_OP = {"sequence": 0, "sequence_imm": 1, "sequence_struct": 2, "within": 3, "within_struct": 4, "any": 5, "and": 6}

DOC_STATUS = {
    0: "ok", 1: "term events not fed in ascending order", 2: "working set of the document exceeds the arena",
    3: "pattern with too many identical key events defined", 4: "internal: encountered past trigger with follow",
    5: "term event out of range", 6: "illegal free of event data reference",
}


class PatternError(RuntimeError):
    pass


class MatchBatch:
    """Results of a batch of documents (host copies)."""

    def __init__(self, results, items, doc_offsets, stats, status, result_format=None, item_format=None):
        self.result_format = result_format  # (n,) u32 format handle per result, None without format strings
        self.item_format = item_format      # (m, 2) u32 {format handle, nsub} per item (see resultformat.py)
        self.results = results          # (n, 9) u32: handle, ordpos, ordend, origseg, origpos, origendseg, origend, item_begin, item_count
        self.items = items              # (m, 7) u32: variable, ordpos, ordend, origseg, origpos, origendseg, origend
        self.doc_offsets = doc_offsets  # (ndocs+1,) u64
        self.stats = stats              # (ndocs, 4) u64
        self.status = status            # (ndocs,) i32

    def doc(self, i):
        return self.results[self.doc_offsets[i]:self.doc_offsets[i + 1]]


def _formats_of(b):
    if not b.result_format:
        return None, None
    rf = np.ctypeslib.as_array(b.result_format, shape=(b.nresults + 1,))[:b.nresults].copy()
    itf = np.ctypeslib.as_array(b.item_format, shape=(b.nitems * 2 + 2,))[:b.nitems * 2].reshape(-1, 2).copy()
    return rf, itf


class PatternMatcherContext:
    """PatternMatcherContextInterface (src/patternMatcher.cpp:107-341) + batch mode."""

    def __init__(self, instance, device=0):
        self._L = capi.lib()
        self._inst = instance  # keeps the instance alive: a context borrows its tables
        self._h = self._L.sp_matcher_ctx_create(instance._h, device)
        if not self._h:
            raise PatternError("failed to create pattern match context: " + self._L.sp_matcher_last_error(instance._h).decode())

    def __del__(self):
        try:
            if self._h:
                self._L.sp_matcher_ctx_free(self._h)
        except Exception:
            pass

    def _err(self):
        return self._L.sp_matcher_ctx_last_error(self._h).decode()

    def setArena(self, max_rules=0, max_triggers=0, bucket_capacity=0, max_items=0, max_follow=0):
        self._L.sp_matcher_ctx_set_arena(self._h, max_rules, max_triggers, bucket_capacity, max_items, max_follow)

    # -- single document mode
    def putInput(self, lexem_id, ordpos, origpos, origsize, origseg=0):
        lx = SpLexem(lexem_id, ordpos, origpos, origsize)
        seg = ctypes.c_uint32(origseg)
        rc = self._L.sp_matcher_ctx_put_input(self._h, ctypes.addressof(lx), ctypes.addressof(seg) if origseg else None, 1)
        if rc != 0:
            raise PatternError("failed to feed input to pattern matcher: " + self._err())

    def fetchResults(self):
        res = ctypes.POINTER(SpResult)()
        items = ctypes.POINTER(SpResultItem)()
        nres = ctypes.c_size_t()
        nitems = ctypes.c_size_t()
        rc = self._L.sp_matcher_ctx_fetch_results(self._h, ctypes.byref(res), ctypes.byref(nres), ctypes.byref(items), ctypes.byref(nitems))
        if rc != 0:
            raise PatternError("failed to fetch pattern match result: " + self._err())
        try:
            r = np.ctypeslib.as_array(ctypes.cast(res, ctypes.POINTER(ctypes.c_uint32)), shape=(nres.value * 9 + 1,))[:nres.value * 9].reshape(-1, 9).copy()
            it = np.ctypeslib.as_array(ctypes.cast(items, ctypes.POINTER(ctypes.c_uint32)), shape=(nitems.value * 7 + 1,))[:nitems.value * 7].reshape(-1, 7).copy()
        finally:
            self._L.sp_free(res)
            self._L.sp_free(items)
        return r, it

    def fetchFormats(self, nresults, nitems):
        """format handles of the results / items of the last fetchResults() (None, None without format strings)."""
        rf = ctypes.POINTER(ctypes.c_uint32)()
        itf = ctypes.POINTER(ctypes.c_uint32)()
        self._L.sp_matcher_ctx_fetch_formats(self._h, ctypes.byref(rf), ctypes.byref(itf))
        if not rf and not self._inst.formatCount():
            return None, None
        r = np.ctypeslib.as_array(rf, shape=(nresults + 1,))[:nresults].copy() if nresults else np.zeros(0, np.uint32)
        i = np.ctypeslib.as_array(itf, shape=(nitems * 2 + 2,))[:nitems * 2].reshape(-1, 2).copy() if nitems else np.zeros((0, 2), np.uint32)
        return r, i

    def getStatistics(self):
        st = capi.SpMatcherStats()
        self._L.sp_matcher_ctx_statistics(self._h, ctypes.byref(st))
        return {n: getattr(st, n) for n, _ in capi.SpMatcherStats._fields_}

    def reset(self):
        self._L.sp_matcher_ctx_reset(self._h)

    # -- batch mode (host buffers)
    def matchDocs(self, lexems, doc_offsets, origseg=None, check=True):
        """lexems: (n,4) u32 [id, ordpos, origpos, origsize]; doc_offsets: (ndocs+1,) u64."""
        lexems = np.ascontiguousarray(lexems, dtype=np.uint32).reshape(-1, 4)
        doc_offsets = np.ascontiguousarray(doc_offsets, dtype=np.uint64)
        ndocs = len(doc_offsets) - 1
        segp = None
        if origseg is not None:
            origseg = np.ascontiguousarray(origseg, dtype=np.uint32)
            segp = origseg.ctypes.data
        b = capi.SpMatchBatch()
        rc = self._L.sp_matcher_ctx_match_docs(self._h, lexems.ctypes.data, segp, doc_offsets.ctypes.data, ndocs, ctypes.byref(b))
        try:
            if rc != 0 and (rc != -5 or check):
                raise PatternError("batch match failed (%d): %s" % (rc, self._err()))
            u32p = ctypes.POINTER(ctypes.c_uint32)
            res = np.ctypeslib.as_array(ctypes.cast(b.results, u32p), shape=(b.nresults * 9 + 1,))[:b.nresults * 9].reshape(-1, 9).copy()
            items = np.ctypeslib.as_array(ctypes.cast(b.items, u32p), shape=(b.nitems * 7 + 1,))[:b.nitems * 7].reshape(-1, 7).copy()
            offs = np.ctypeslib.as_array(b.doc_result_offsets, shape=(ndocs + 1,)).copy()
            stats = np.ctypeslib.as_array(b.doc_stats, shape=(ndocs * 4 + 1,))[:ndocs * 4].reshape(-1, 4).copy()
            status = np.ctypeslib.as_array(b.doc_status, shape=(ndocs + 1,))[:ndocs].copy()
            rfmt, ifmt = _formats_of(b)
        finally:
            self._L.sp_match_batch_free(ctypes.byref(b))
        return MatchBatch(res, items, offs, stats, status, rfmt, ifmt)

    # -- batch mode (device-resident buffers; pointers are raw device addresses, e.g. torch data_ptr())
    def matchDocsDevice(self, d_lexems, d_doc_offsets, ndocs, nlexems, stream=0, d_origseg=0):
        out = capi.SpMatchDeviceBatch()
        rc = self._L.sp_matcher_ctx_match_docs_device(self._h, d_lexems, d_origseg or None, d_doc_offsets, ndocs, nlexems, stream or None, ctypes.byref(out))
        if rc != 0:
            raise PatternError("device batch match failed (%d): %s" % (rc, self._err()))
        return out

    def matchLexedDevice(self, d_lexems, d_doc_ranges, ndocs, nlexems_hint, stream=0):
        """consume the device output of PatternLexerContext.matchDocsDevice in place (fused pipeline)."""
        out = capi.SpMatchDeviceBatch()
        rc = self._L.sp_matcher_ctx_match_lexed_device(self._h, d_lexems, d_doc_ranges, ndocs, nlexems_hint, stream or None, ctypes.byref(out))
        if rc != 0:
            raise PatternError("device batch match failed (%d): %s" % (rc, self._err()))
        return out

    def batchFetch(self, first_doc=None, ndocs=None):
        """host copy of the last device batch, grouped by document; with (first_doc, ndocs) only of
        that range of documents."""
        b = capi.SpMatchBatch()
        if first_doc is None:
            rc = self._L.sp_matcher_ctx_batch_fetch(self._h, ctypes.byref(b))
        else:
            rc = self._L.sp_matcher_ctx_batch_fetch_docs(self._h, first_doc, ndocs, ctypes.byref(b))
        try:
            if rc != 0:
                raise PatternError("fetching the batch failed (%d): %s" % (rc, self._err()))
            ndocs = b.ndocs
            u32p = ctypes.POINTER(ctypes.c_uint32)
            res = np.ctypeslib.as_array(ctypes.cast(b.results, u32p), shape=(b.nresults * 9 + 1,))[:b.nresults * 9].reshape(-1, 9).copy()
            items = np.ctypeslib.as_array(ctypes.cast(b.items, u32p), shape=(b.nitems * 7 + 1,))[:b.nitems * 7].reshape(-1, 7).copy()
            offs = np.ctypeslib.as_array(b.doc_result_offsets, shape=(ndocs + 1,)).copy()
            stats = np.ctypeslib.as_array(b.doc_stats, shape=(ndocs * 4 + 1,))[:ndocs * 4].reshape(-1, 4).copy()
            status = np.ctypeslib.as_array(b.doc_status, shape=(ndocs + 1,))[:ndocs].copy()
            rfmt, ifmt = _formats_of(b)
        finally:
            self._L.sp_match_batch_free(ctypes.byref(b))
        return MatchBatch(res, items, offs, stats, status, rfmt, ifmt)

    def batchCounters(self):
        arr = (ctypes.c_uint64 * 8)()
        rc = self._L.sp_matcher_ctx_batch_counters(self._h, arr)
        if rc != 0:
            raise PatternError("reading batch counters failed: " + self._err())
        return {"results": arr[0], "items": arr[1], "events": arr[2], "failed_docs": arr[3], "handed_over": arr[4], "prof": [arr[4], arr[5], arr[6], arr[7]]}

    def batchStatus(self, ndocs):
        st = np.zeros(ndocs, np.int32)
        rc = self._L.sp_matcher_ctx_batch_status(self._h, st.ctypes.data, ndocs)
        if rc != 0:
            raise PatternError("reading batch status failed: " + self._err())
        return st

    def growArena(self):
        return self._L.sp_matcher_ctx_grow_arena(self._h) == 0

    def reserveOutput(self, results, items):
        self._L.sp_matcher_ctx_reserve_output(self._h, results, items)

    def lastKernelMs(self):
        return self._L.sp_matcher_ctx_last_kernel_ms(self._h)

    def kernelKind(self):
        """0 general automaton kernel, 1 LDS-resident kernel (flat rule sets), 2 join prototype (SPA_L2_JOIN=1)"""
        return self._L.sp_matcher_ctx_kernel_kind(self._h)

    def kernelName(self):
        """name of the kernel that does the work of this context's batches (what rocprofv3 lists)"""
        return self._L.sp_matcher_ctx_kernel_name(self._h).decode()


def _serialize(L, fn, handle, err):
    blob = ctypes.c_void_p()
    size = ctypes.c_size_t()
    if fn(handle, ctypes.byref(blob), ctypes.byref(size)) != 0:
        raise PatternError("serialisation failed: " + err())
    try:
        return ctypes.string_at(blob, size.value)
    finally:
        L.sp_free(blob)


class PatternMatcherInstance:
    """PatternMatcherInstanceInterface (src/patternMatcher.cpp:345-733)."""

    def __init__(self, _handle=None):
        self._L = capi.lib()
        self._h = _handle or self._L.sp_matcher_create()
        if not self._h:
            raise PatternError("failed to create pattern matcher instance")

    def serialize(self):
        """the rule set (tables, names, format strings, options) as bytes: SURVEY.md 8(f).4"""
        return _serialize(self._L, self._L.sp_matcher_serialize, self._h, lambda: self._L.sp_matcher_last_error(self._h).decode())

    @classmethod
    def deserialize(cls, blob):
        L = capi.lib()
        err = ctypes.create_string_buffer(256)
        h = L.sp_matcher_deserialize(blob, len(blob), err, 256)
        if not h:
            raise PatternError("cannot load the rule set: " + err.value.decode())
        return cls(_handle=h)

    def __del__(self):
        try:
            if self._h:
                self._L.sp_matcher_free(self._h)
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != 0:
            raise PatternError("%s: %s" % (what, self._L.sp_matcher_last_error(self._h).decode()))

    def defineTermFrequency(self, termid, df):
        self._chk(self._L.sp_matcher_define_term_frequency(self._h, termid, df), "failed to define term frequency")

    def pushTerm(self, termid):
        self._chk(self._L.sp_matcher_push_term(self._h, termid), "failed to push term on the pattern match expression stack")

    def pushExpression(self, joinop, argc, range_, cardinality=0):
        op = JOIN_OP[joinop] if isinstance(joinop, str) else int(joinop)
        self._chk(self._L.sp_matcher_push_expression(self._h, op, argc, range_, cardinality), "failed to push expression on the pattern match expression stack")

    def pushPattern(self, name):
        self._chk(self._L.sp_matcher_push_pattern(self._h, name.encode()), "failed to push pattern reference on the pattern match expression stack")

    def attachVariable(self, name):
        self._chk(self._L.sp_matcher_attach_variable(self._h, name.encode()), "failed to attach variable to top element of the pattern match expression stack")

    def definePattern(self, name, formatstring="", visible=True):
        self._chk(self._L.sp_matcher_define_pattern(self._h, name.encode(), formatstring.encode(), int(visible)), "failed to close pattern definition on the pattern match expression stack")

    def defineOption(self, name, value=0.0):
        self._chk(self._L.sp_matcher_define_option(self._h, name.encode(), value), "failed to define pattern matching automaton option")

    def compile(self):
        self._chk(self._L.sp_matcher_compile(self._h), "failed to compile (optimize) pattern matching automaton")
        return True

    def createContext(self, device=0):
        return PatternMatcherContext(self, device)

    def patternId(self, name):
        return self._L.sp_matcher_pattern_id(self._h, name.encode())

    def patternName(self, handle):
        s = self._L.sp_matcher_pattern_name(self._h, handle)
        return s.decode() if s else None

    def variableId(self, name):
        return self._L.sp_matcher_variable_id(self._h, name.encode())

    def variableName(self, vid):
        s = self._L.sp_matcher_variable_name(self._h, vid)
        return s.decode() if s else None

    def formatCount(self):
        return int(self._L.sp_matcher_format_count(self._h))

    def formatString(self, handle):
        s = self._L.sp_matcher_format_string(self._h, handle)
        return s.decode() if s else None

    def fastTier(self):
        """(True, "") when the rule set is flat and runs on the LDS-resident kernel, else (False, reason)."""
        buf = ctypes.create_string_buffer(256)
        ok = self._L.sp_matcher_fast_tier(self._h, buf, 256)
        return bool(ok), buf.value.decode()

    def dumpTable(self):
        p = ctypes.POINTER(ctypes.c_uint32)()
        n = self._L.sp_matcher_dump_table(self._h, ctypes.byref(p))
        arr = np.ctypeslib.as_array(p, shape=(n,)).copy() if n else np.zeros(0, np.uint32)
        self._L.sp_free(p)
        return arr

    def name(self):
        return "std"


class PatternMatcher:
    """PatternMatcherInterface (src/patternMatcher.hpp:31-35)."""

    def getCompileOptionNames(self):
        return ["stopwordOccurrenceFactor", "weightFactor", "maxRange", "exclusive", "maxResultSize"]  # src/patternMatcher.cpp:707-716

    def createInstance(self):
        return PatternMatcherInstance()

    def name(self):
        return "std"


POSITION_BIND = {"content": 0, "successor": 1, "predecessor": 2, "unique": 3}


class LexBatch:
    def __init__(self, lexems, doc_offsets, status):
        self.lexems = lexems            # (n,4) u32: id, ordpos, origpos, origsize
        self.doc_offsets = doc_offsets  # (ndocs+1,) u64
        self.status = status            # (ndocs,) i32

    def doc(self, i):
        return self.lexems[self.doc_offsets[i]:self.doc_offsets[i + 1]]


class PatternLexerContext:
    """PatternLexerContextInterface (src/patternLexer.cpp:681-959) + batch mode."""

    def __init__(self, instance, device=0):
        self._L = capi.lib()
        self._inst = instance
        self._h = self._L.sp_lexer_ctx_create(instance._h, device)
        if not self._h:
            raise PatternError("failed to create term match context: " + self._L.sp_lexer_last_error(instance._h).decode())

    def __del__(self):
        try:
            if self._h:
                self._L.sp_lexer_ctx_free(self._h)
        except Exception:
            pass

    def _err(self):
        return self._L.sp_lexer_ctx_last_error(self._h).decode(errors="replace")

    def match(self, src):
        """std::vector<PatternLexem> match(const char* src, size_t len) (:858)."""
        buf = bytes(src)
        lex = ctypes.POINTER(SpLexem)()
        n = ctypes.c_size_t()
        rc = self._L.sp_lexer_ctx_match(self._h, buf, len(buf), ctypes.byref(lex), ctypes.byref(n))
        if rc != 0:
            raise PatternError("failed to run pattern matching terms with regular expressions: " + self._err())
        try:
            arr = np.ctypeslib.as_array(ctypes.cast(lex, ctypes.POINTER(ctypes.c_uint32)), shape=(n.value * 4 + 1,))[:n.value * 4].reshape(-1, 4).copy()
        finally:
            self._L.sp_free(lex)
        return arr

    def reset(self):
        self._L.sp_lexer_ctx_reset(self._h)

    def matchDocs(self, text, doc_offsets, check=True):
        buf = bytes(text)
        doc_offsets = np.ascontiguousarray(doc_offsets, dtype=np.uint64)
        ndocs = len(doc_offsets) - 1
        b = capi.SpLexBatch()
        rc = self._L.sp_lexer_ctx_match_docs(self._h, buf, doc_offsets.ctypes.data, ndocs, ctypes.byref(b))
        try:
            if rc != 0 and (rc != -5 or check):
                raise PatternError("batch lexer run failed (%d): %s" % (rc, self._err()))
            lex = np.ctypeslib.as_array(ctypes.cast(b.lexems, ctypes.POINTER(ctypes.c_uint32)), shape=(b.nlexems * 4 + 1,))[:b.nlexems * 4].reshape(-1, 4).copy()
            offs = np.ctypeslib.as_array(b.doc_lexem_offsets, shape=(ndocs + 1,)).copy()
            status = np.ctypeslib.as_array(b.doc_status, shape=(ndocs + 1,))[:ndocs].copy()
        finally:
            self._L.sp_lex_batch_free(ctypes.byref(b))
        return LexBatch(lex, offs, status)

    def matchDocsDevice(self, d_text, d_doc_offsets, ndocs, nbytes, stream=0):
        out = capi.SpLexDeviceBatch()
        rc = self._L.sp_lexer_ctx_match_docs_device(self._h, d_text, d_doc_offsets, ndocs, nbytes, stream or None, ctypes.byref(out))
        if rc != 0:
            raise PatternError("device batch lexer run failed (%d): %s" % (rc, self._err()))
        return out

    def batchFetch(self, first_doc, ndocs):
        """host copy of the lexems of the documents [first_doc, first_doc+ndocs) of the last device batch."""
        b = capi.SpLexBatch()
        rc = self._L.sp_lexer_ctx_batch_fetch_docs(self._h, first_doc, ndocs, ctypes.byref(b))
        try:
            if rc != 0:
                raise PatternError("fetching the lexer batch failed (%d): %s" % (rc, self._err()))
            lex = np.ctypeslib.as_array(ctypes.cast(b.lexems, ctypes.POINTER(ctypes.c_uint32)), shape=(b.nlexems * 4 + 1,))[:b.nlexems * 4].reshape(-1, 4).copy()
            offs = np.ctypeslib.as_array(b.doc_lexem_offsets, shape=(ndocs + 1,)).copy()
            status = np.ctypeslib.as_array(b.doc_status, shape=(ndocs + 1,))[:ndocs].copy()
        finally:
            self._L.sp_lex_batch_free(ctypes.byref(b))
        return LexBatch(lex, offs, status)

    def batchCounters(self):
        arr = (ctypes.c_uint64 * 8)()
        if self._L.sp_lexer_ctx_batch_counters(self._h, arr) != 0:
            raise PatternError("reading batch counters failed: " + self._err())
        return {"lexems": arr[0], "bytes": arr[1], "raw_reports": arr[2], "failed_docs": arr[3], "scan_units": arr[4], "rescanned_docs": arr[5],
                "word_reports": arr[6], "prof": [arr[4], arr[5], arr[6], arr[7]]}

    def batchStatus(self, ndocs):
        st = np.zeros(ndocs, np.int32)
        if self._L.sp_lexer_ctx_batch_status(self._h, st.ctypes.data, ndocs) != 0:
            raise PatternError("reading batch status failed: " + self._err())
        return st

    def lastKernelMs(self):
        return self._L.sp_lexer_ctx_last_kernel_ms(self._h)

    def lastKernelMsSplit(self):
        """(scan kernel ms, post-processing kernel ms) of the last launch"""
        a, b = ctypes.c_double(-1.0), ctypes.c_double(-1.0)
        if self._L.sp_lexer_ctx_last_kernel_ms_split(self._h, ctypes.byref(a), ctypes.byref(b)) != 0:
            raise PatternError("no timed launch")
        return a.value, b.value

    def lastKernelMsSplit3(self):
        """(scan kernel ms, words kernel ms, post-processing kernel ms) of the last launch"""
        a, b, d = ctypes.c_double(-1.0), ctypes.c_double(-1.0), ctypes.c_double(-1.0)
        if self._L.sp_lexer_ctx_last_kernel_ms_split3(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(d)) != 0:
            raise PatternError("no timed launch")
        return a.value, b.value, d.value

    def wordsKernelName(self):
        """the instance of the words kernel the last launch went through"""
        return self._L.sp_lexer_ctx_words_kernel_name(self._h).decode()

    def scanKernelName(self):
        """the scan kernel the last launch went through"""
        return self._L.sp_lexer_ctx_scan_kernel_name(self._h).decode()

    def reserveOutput(self, lexems):
        self._L.sp_lexer_ctx_reserve_output(self._h, lexems)

    def growArena(self):
        return self._L.sp_lexer_ctx_grow_arena(self._h) == 0


class PatternLexerInstance:
    """PatternLexerInstanceInterface (src/patternLexer.cpp:961-1151)."""

    def __init__(self, _handle=None):
        self._L = capi.lib()
        self._h = _handle or self._L.sp_lexer_create()
        if not self._h:
            raise PatternError("failed to create term match instance")

    def serialize(self):
        """the compiled lexer (automaton, literal and symbol tables, names) as bytes: SURVEY.md 8(f).4"""
        return _serialize(self._L, self._L.sp_lexer_serialize, self._h, lambda: self._L.sp_lexer_last_error(self._h).decode(errors="replace"))

    @classmethod
    def deserialize(cls, blob):
        L = capi.lib()
        err = ctypes.create_string_buffer(256)
        h = L.sp_lexer_deserialize(blob, len(blob), err, 256)
        if not h:
            raise PatternError("cannot load the lexer: " + err.value.decode())
        return cls(_handle=h)

    def __del__(self):
        try:
            if self._h:
                self._L.sp_lexer_free(self._h)
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != 0:
            raise PatternError("%s: %s" % (what, self._L.sp_lexer_last_error(self._h).decode(errors="replace")))

    @staticmethod
    def _b(s):
        return s if isinstance(s, bytes) else s.encode()

    def defineLexemName(self, id_, name):
        self._chk(self._L.sp_lexer_define_lexem_name(self._h, id_, self._b(name)), "failed to assign lexem name to lexem or symbol identifier")

    def getLexemName(self, id_):
        s = self._L.sp_lexer_get_lexem_name(self._h, id_)
        return s.decode() if s else None

    def defineLexem(self, id_, expression, resultIndex=0, level=0, posbind="content"):
        pb = POSITION_BIND[posbind] if isinstance(posbind, str) else int(posbind)
        self._chk(self._L.sp_lexer_define_lexem(self._h, id_, self._b(expression), resultIndex, level, pb), "failed to define term match regular expression pattern")

    def defineSymbol(self, symbolid, patternid, name):
        self._chk(self._L.sp_lexer_define_symbol(self._h, symbolid, patternid, self._b(name)), "failed to define regular expression pattern symbol")

    def getSymbol(self, patternid, name):
        return self._L.sp_lexer_get_symbol(self._h, patternid, self._b(name))

    def defineOption(self, name, value=0.0):
        self._chk(self._L.sp_lexer_define_option(self._h, name.encode(), value), "define option failed for pattern lexer")

    def compile(self):
        self._chk(self._L.sp_lexer_compile(self._h), "failed to compile regular expression patterns")
        return True

    def createContext(self, device=0):
        return PatternLexerContext(self, device)

    def dumpTables(self):
        p = ctypes.POINTER(ctypes.c_uint64)()
        n = self._L.sp_lexer_dump_tables(self._h, ctypes.byref(p))
        arr = np.ctypeslib.as_array(p, shape=(n,)).copy() if n else np.zeros(0, np.uint64)
        self._L.sp_free(p)
        return arr

    def name(self):
        return "std"


class PatternLexer:
    """PatternLexerInterface (src/patternLexer.hpp:22-41)."""

    def getCompileOptionNames(self):
        return ["CASELESS", "DOTALL", "MULTILINE", "ALLOWEMPTY", "UCP"]  # src/patternLexer.cpp:1154-1163

    def createInstance(self):
        return PatternLexerInstance()

    def name(self):
        return "std"


def device_count():
    return capi.lib().sp_device_count()
