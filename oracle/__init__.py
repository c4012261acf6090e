"""TEST INFRASTRUCTURE ONLY -- ctypes loader for the CPU oracle (oracle/_build/liboracle.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product package (struspattern_amd) never does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

OP = {"sequence": 0, "sequence_imm": 1, "sequence_struct": 2, "within": 3, "within_struct": 4, "any": 5, "and": 6}


def build(force=False):
    """Compile the oracle with g++ (a few seconds)."""
    so = os.path.join(_HERE, "_build", "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".cpp", ".hpp"))]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return so


class _L2Out(ctypes.Structure):
    _fields_ = [
        ("nresults", ctypes.c_uint64),
        ("nitems", ctypes.c_uint64),
        ("results", ctypes.POINTER(ctypes.c_uint32)),
        ("items", ctypes.POINTER(ctypes.c_uint32)),
        ("doc_result_offsets", ctypes.POINTER(ctypes.c_uint64)),
        ("doc_stats", ctypes.POINTER(ctypes.c_uint64)),
        ("doc_status", ctypes.POINTER(ctypes.c_int32)),
        ("result_format", ctypes.POINTER(ctypes.c_uint32)),
        ("item_format", ctypes.POINTER(ctypes.c_uint32)),
    ]


def lib():
    global _LIB
    if _LIB is None:
        so = build()
        L = ctypes.CDLL(so)
        L.orc_l2_new.restype = ctypes.c_void_p
        L.orc_l2_free.argtypes = [ctypes.c_void_p]
        L.orc_l2_last_error.restype = ctypes.c_char_p
        L.orc_l2_last_error.argtypes = [ctypes.c_void_p]
        L.orc_l2_define_term_frequency.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_double]
        L.orc_l2_push_term.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
        L.orc_l2_push_expression.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]
        L.orc_l2_push_pattern.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
        L.orc_l2_attach_variable.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
        L.orc_l2_define_pattern.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int]
        L.orc_l2_define_option.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_double]
        L.orc_l2_compile.argtypes = [ctypes.c_void_p]
        L.orc_l2_pattern_id.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
        L.orc_l2_pattern_id.restype = ctypes.c_uint32
        L.orc_l2_variable_id.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
        L.orc_l2_variable_id.restype = ctypes.c_uint32
        L.orc_l2_dump_table.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.POINTER(ctypes.c_uint32))]
        L.orc_l2_dump_table.restype = ctypes.c_uint64
        L.orc_l2_run_docs.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.POINTER(_L2Out)]
        L.orc_l2_free_out.argtypes = [ctypes.POINTER(_L2Out)]
        L.orc_free.argtypes = [ctypes.c_void_p]
        _LIB = L
    return _LIB


class OracleError(RuntimeError):
    pass


class L2Results:
    """Per-document results of a batch run (numpy views copied out of the C buffers)."""

    def __init__(self, results, items, doc_offsets, stats, status):
        self.results = results          # (n, 9) u32: handle, sord, eord, sseg, spos, eseg, epos, item_begin, item_count
        self.items = items              # (m, 7) u32: variable, sord, eord, sseg, spos, eseg, epos
        self.doc_offsets = doc_offsets  # (ndocs+1,) u64
        self.stats = stats              # (ndocs, 4) u64
        self.status = status            # (ndocs,) i32

    def doc(self, i):
        return self.results[self.doc_offsets[i]:self.doc_offsets[i + 1]]


class L2Matcher:
    """Mirror of PatternMatcherInstanceInterface (patternMatcher.cpp:361-680) on the CPU oracle."""

    def __init__(self):
        self._L = lib()
        self._h = self._L.orc_l2_new()

    def __del__(self):
        try:
            self._L.orc_l2_free(self._h)
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise OracleError(self._L.orc_l2_last_error(self._h).decode())

    def defineTermFrequency(self, termid, df):
        self._chk(self._L.orc_l2_define_term_frequency(self._h, termid, df))

    def pushTerm(self, termid):
        self._chk(self._L.orc_l2_push_term(self._h, termid))

    def pushExpression(self, op, argc, range_, cardinality=0):
        self._chk(self._L.orc_l2_push_expression(self._h, OP[op] if isinstance(op, str) else op, argc, range_, cardinality))

    def pushPattern(self, name):
        self._chk(self._L.orc_l2_push_pattern(self._h, name.encode()))

    def attachVariable(self, name):
        self._chk(self._L.orc_l2_attach_variable(self._h, name.encode()))

    def definePattern(self, name, formatstring="", visible=True):
        self._chk(self._L.orc_l2_define_pattern(self._h, name.encode(), formatstring.encode(), int(visible)))

    def defineOption(self, name, value=0.0):
        self._chk(self._L.orc_l2_define_option(self._h, name.encode(), value))

    def compile(self):
        self._chk(self._L.orc_l2_compile(self._h))

    def patternId(self, name):
        return self._L.orc_l2_pattern_id(self._h, name.encode())

    def variableId(self, name):
        return self._L.orc_l2_variable_id(self._h, name.encode())

    def dumpTable(self):
        p = ctypes.POINTER(ctypes.c_uint32)()
        n = self._L.orc_l2_dump_table(self._h, ctypes.byref(p))
        arr = np.ctypeslib.as_array(p, shape=(n,)).copy() if n else np.zeros(0, np.uint32)
        self._L.orc_free(p)
        return arr

    def run(self, lexems, doc_offsets, nthreads=1):
        """lexems: (n,5) u32 [id, ordpos, origseg, origpos, origsize]; doc_offsets: (ndocs+1,) u64."""
        lexems = np.ascontiguousarray(lexems, dtype=np.uint32).reshape(-1, 5)
        doc_offsets = np.ascontiguousarray(doc_offsets, dtype=np.uint64)
        ndocs = len(doc_offsets) - 1
        out = _L2Out()
        rc = self._L.orc_l2_run_docs(self._h, lexems.ctypes.data, doc_offsets.ctypes.data, ndocs, nthreads, ctypes.byref(out))
        try:
            res = np.ctypeslib.as_array(out.results, shape=(out.nresults * 9 + 1,))[:out.nresults * 9].reshape(-1, 9).copy()
            items = np.ctypeslib.as_array(out.items, shape=(out.nitems * 7 + 1,))[:out.nitems * 7].reshape(-1, 7).copy()
            offs = np.ctypeslib.as_array(out.doc_result_offsets, shape=(ndocs + 1,)).copy()
            stats = np.ctypeslib.as_array(out.doc_stats, shape=(ndocs * 4 + 1,))[:ndocs * 4].reshape(-1, 4).copy()
            status = np.ctypeslib.as_array(out.doc_status, shape=(ndocs + 1,))[:ndocs].copy()
            rfmt = ifmt = None
            if out.result_format:
                rfmt = np.ctypeslib.as_array(out.result_format, shape=(out.nresults + 1,))[:out.nresults].copy()
                ifmt = np.ctypeslib.as_array(out.item_format, shape=(out.nitems * 2 + 2,))[:out.nitems * 2].reshape(-1, 2).copy()
        finally:
            self._L.orc_l2_free_out(ctypes.byref(out))
        if rc != 0:
            raise OracleError(self._L.orc_l2_last_error(self._h).decode())
        r = L2Results(res, items, offs, stats, status)
        r.result_format, r.item_format = rfmt, ifmt
        return r


POSBIND = {"content": 0, "successor": 1, "predecessor": 2, "unique": 3}


def _lib_l1():
    L = lib()
    if not getattr(L, "_l1_ready", False):
        L.orc_l1_new.restype = ctypes.c_void_p
        L.orc_l1_free.argtypes = [ctypes.c_void_p]
        L.orc_l1_last_error.restype = ctypes.c_char_p
        L.orc_l1_last_error.argtypes = [ctypes.c_void_p]
        L.orc_l1_define_lexem.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int]
        L.orc_l1_define_symbol.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_char_p]
        L.orc_l1_define_option.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_double]
        L.orc_l1_compile.argtypes = [ctypes.c_void_p]
        L.orc_l1_get_symbol.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_char_p]
        L.orc_l1_get_symbol.restype = ctypes.c_uint32
        L.orc_l1_match_docs.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_int,
                                        ctypes.POINTER(ctypes.POINTER(ctypes.c_uint32)), ctypes.POINTER(ctypes.POINTER(ctypes.c_uint64))]
        L._l1_ready = True
    return L


class L1Lexer:
    """Mirror of PatternLexerInstanceInterface (patternLexer.cpp:961-1151) on the CPU oracle."""

    def __init__(self):
        self._L = _lib_l1()
        self._h = self._L.orc_l1_new()

    def __del__(self):
        try:
            self._L.orc_l1_free(self._h)
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise OracleError(self._L.orc_l1_last_error(self._h).decode(errors="replace"))

    def defineLexem(self, id_, expression, resultIndex=0, level=0, posbind="content"):
        expr = expression if isinstance(expression, bytes) else expression.encode()
        pb = POSBIND[posbind] if isinstance(posbind, str) else int(posbind)
        self._chk(self._L.orc_l1_define_lexem(self._h, id_, expr, resultIndex, level, pb))

    def defineSymbol(self, symbolid, patternid, name):
        nm = name if isinstance(name, bytes) else name.encode()
        self._chk(self._L.orc_l1_define_symbol(self._h, symbolid, patternid, nm))

    def getSymbol(self, patternid, name):
        nm = name if isinstance(name, bytes) else name.encode()
        return self._L.orc_l1_get_symbol(self._h, patternid, nm)

    def defineOption(self, name, value=0.0):
        self._chk(self._L.orc_l1_define_option(self._h, name.encode(), value))

    def compile(self):
        self._chk(self._L.orc_l1_compile(self._h))
        return True

    def matchDocs(self, text, doc_offsets, nthreads=1, raw=False):
        """text: bytes of all documents concatenated; doc_offsets: (ndocs+1,) u64.
        Returns (lexems (n,4) u32 [id, ordpos, origpos, origsize], lexem_doc_offsets (ndocs+1,) u64).
        raw=True returns the report stream [patternidx, from, to, 0] that feeds the event handler."""
        doc_offsets = np.ascontiguousarray(doc_offsets, dtype=np.uint64)
        ndocs = len(doc_offsets) - 1
        buf = bytes(text)
        pl = ctypes.POINTER(ctypes.c_uint32)()
        po = ctypes.POINTER(ctypes.c_uint64)()
        rc = self._L.orc_l1_match_docs(self._h, buf, doc_offsets.ctypes.data, ndocs, nthreads, int(raw), ctypes.byref(pl), ctypes.byref(po))
        try:
            offs = np.ctypeslib.as_array(po, shape=(ndocs + 1,)).copy()
            n = int(offs[-1])
            lex = np.ctypeslib.as_array(pl, shape=(n * 4 + 1,))[:n * 4].reshape(-1, 4).copy()
        finally:
            self._L.orc_free(pl)
            self._L.orc_free(po)
        if rc != 0:
            raise OracleError(self._L.orc_l1_last_error(self._h).decode(errors="replace"))
        return lex, offs

    def match(self, text):
        lex, _ = self.matchDocs(text, [0, len(text)])
        return lex
