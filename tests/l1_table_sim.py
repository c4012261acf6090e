"""Test utility: a tiny pure-Python interpreter of the product's compiled lexer tables
(sp_lexer_dump_tables).  It executes exactly the recurrences the HIP kernel executes -- forward
scan with per-word bit masks, backward start-of-match resolution -- with Python integers, so the
host regex compiler can be checked against the oracle on the CPU, without a GPU.  Test code only:
nothing in the product imports this."""
import numpy as np

M64 = (1 << 64) - 1
CTX_EDGE = 3


class Tables:
    def __init__(self, dump):
        d = [int(x) for x in dump]
        self.nofPasses, self.nofClasses, self.maxEx, self.nofPatterns, self.nofPositions, self.nofLiterals = d[0:6]
        self.ucp = bool(d[7])
        p = 8
        nsym = 320 if self.ucp else 256     # UCP: [256..319] = continuation bytes 80..BF of a word character
        self.byteClass = d[p:p + nsym]; p += nsym
        self.classCtx = d[p:p + self.nofClasses]; p += self.nofClasses
        P, C = self.nofPasses, self.nofClasses
        E = max(self.maxEx, 1)

        def take(n):
            nonlocal p
            v = d[p:p + n]
            p += n
            return v
        self.charMask = take(P * C * 64)
        self.startMask = take(P * 4 * 64)
        self.acceptMask = take(P * 4 * 64)
        self.shiftDst = take(P * 64)
        self.selfLoop = take(P * 64)
        self.exCount = take(P)
        self.exSrc = take(P * E * 64)
        self.exDst = take(P * E * 64)
        self.E = E
        self.patterns = []
        for i in range(self.nofPatterns):
            pid, word, lb, pre, suf, mask, defidx = take(7)
            self.patterns.append(dict(id=pid, word=word, levelBind=lb, prefixLen=pre, suffixLen=suf, mask=mask, defIndex=defidx))
        self.literals = {}
        for i in range(self.nofLiterals):
            ln, pc = take(2)
            word = bytes(take(ln))
            self.literals[word] = take(pc)
        nnull = take(1)[0]
        self.nullable = [tuple(take(2)) for _ in range(nnull)]      # ALLOWEMPTY: (patterns entry, emptyOk bits)
        nblk = take(1)[0]
        self.cpBlocks = take(nblk)
        npg = take(1)[0]
        self.cpPages = take(npg)
        # word shapes: passes the scan kernel runs, expressions taken as shapes, table entries
        self.scan_passes, self.nof_shapes = take(2)
        self.npasses = self.nofPasses
        self.shapes = {}
        while p < len(d):
            tag, key, pc = take(3)
            self.shapes[(tag, key)] = take(pc)
        assert p == len(d)

    def _lead(self, text, pos):
        """(class by code point or None, length) of the well-formed character starting at pos"""
        b = text[pos]
        if not (self.cpBlocks and 0xC2 <= b <= 0xF4):
            return None, 1
        want = 2 if b <= 0xDF else 3 if b <= 0xEF else 4
        if pos + want <= len(text) and all((text[pos + i] & 0xC0) == 0x80 for i in range(1, want)):
            v = b & (0xFF >> (want + 1))
            for i in range(1, want):
                v = (v << 6) | (text[pos + i] & 0x3F)
            ok = v >= 0x80 if want == 2 else v >= 0x800 if want == 3 else 0x10000 <= v <= 0x10FFFF
            if ok:
                c = self.cpPages[self.cpBlocks[v >> 6] * 64 + (v & 63)]
                if c != 0xFF:
                    return c, want
        return None, 1

    def cls(self, text, pos):
        """class of the byte at pos: by code point for the lead byte of a well-formed multi-byte character when the
        tables have classes by code point; with UCP the twin class for a continuation byte of a word character"""
        b = text[pos]
        c, _ = self._lead(text, pos)
        if c is not None:
            return c
        if self.ucp and 0x80 <= b <= 0xBF:
            for d in range(1, 4):
                if pos - d < 0:
                    break
                l = text[pos - d]
                if 0xC2 <= l <= 0xF4:
                    lc, n = self._lead(text, pos - d)
                    if lc is not None and n > d and self.classCtx[lc] == 0:
                        return self.byteClass[256 + b - 0x80]
                    break
                if (l & 0xC0) != 0x80:
                    break
        return self.byteClass[b]

    def ctx(self, text, pos):
        if pos < 0 or pos >= len(text):
            return CTX_EDGE
        return self.classCtx[self.cls(text, pos)]

    def raw_reports(self, text):
        """[(patternidx 1-based, from, to)] in (to, idx) order."""
        P, C, E = self.nofPasses, self.nofClasses, self.E
        nwords = P * 64
        state = [0] * nwords
        out = []
        prevctx = CTX_EDGE
        for i in range(len(text) + 1):
            ctx = self.ctx(text, i)
            cls = self.cls(text, i) if i < len(text) else 0
            new = list(state)
            for w in range(nwords):
                st = state[w]
                p, ln = divmod(w, 64)
                acc = st & self.acceptMask[(p * 4 + ctx) * 64 + ln]
                if acc:
                    for pi, pat in enumerate(self.patterns):
                        if pat["word"] == w and (acc & pat["mask"]):
                            out.append((pi + 1, self.som(text, pi, i, acc & pat["mask"]), i))
                if i < len(text):
                    nxt = ((st << 1) & M64 & self.shiftDst[w]) | (st & self.selfLoop[w]) | self.startMask[(p * 4 + prevctx) * 64 + ln]
                    for e in range(self.exCount[p]):
                        at = (p * E + e) * 64 + ln
                        if st & self.exSrc[at]:
                            nxt |= self.exDst[at]
                    new[w] = nxt & self.charMask[(p * C + cls) * 64 + ln]
            state = new
            prevctx = ctx
        # whole-word literals: maximal runs of word characters that equal a literal
        i = 0
        n = len(text)
        while i < n:
            if self.ctx(text, i) == 0:
                j = i
                while j < n and self.ctx(text, j) == 0:
                    j += 1
                for pi in self.literals.get(bytes(text[i:j]), []):
                    out.append((pi + 1, i, j))
                i = j
            else:
                i += 1
        # ALLOWEMPTY: the empty match of an expression wherever its empty path holds and nothing longer of it ends
        have = set((r[0], r[2]) for r in out)
        for i in range(len(text) + 1):
            pv, nx = self.ctx(text, i - 1), self.ctx(text, i)
            for pi, ok in self.nullable:
                if (ok >> (pv * 4 + nx)) & 1 and (pi + 1, i) not in have:
                    out.append((pi + 1, i, i))
        out.sort(key=lambda r: (r[2], r[0]))
        # an expression cut into several entries reports once per entry: one report per (definition, end) with the
        # leftmost start, numbered by definition
        merged = []
        for pi, frm, to in out:
            d = self.patterns[pi - 1]["defIndex"] + 1
            if merged and merged[-1][0] == d and merged[-1][2] == to:
                merged[-1] = (d, min(merged[-1][1], frm), to)
            else:
                merged.append((d, frm, to))
        return merged

    def som(self, text, pi, to, R):
        pat = self.patterns[pi]
        w = pat["word"]
        p, ln = divmod(w, 64)
        frm = to
        j = to
        while R and j > 0:
            prevctx = self.ctx(text, j - 2)
            if R & self.startMask[(p * 4 + prevctx) * 64 + ln]:
                frm = j - 1
            if j - 1 == 0:
                break
            Rp = ((R & self.shiftDst[w]) >> 1) | (R & self.selfLoop[w])
            for e in range(self.exCount[p]):
                at = (p * self.E + e) * 64 + ln
                if R & self.exDst[at]:
                    Rp |= self.exSrc[at]
            cls = self.cls(text, j - 2)
            R = Rp & pat["mask"] & self.charMask[(p * self.nofClasses + cls) * 64 + ln]
            j -= 1
        return frm
