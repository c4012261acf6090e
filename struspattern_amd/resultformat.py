"""Result format strings of definePattern.

The reference hands the format string to strusAnalyzer's PatternResultFormat (patternMatcher.cpp:79,
:172-181, :253-262, :561-566), which is NOT part of the reference repository; only its behaviour is
documented (doc/webpage/introduction_struspattern.htm:143-161): `["CHF"]` is a constant, in
`["{currency} {value}"]` every `{variable}` is replaced by the value of the item bound to that
variable -- the value of its own format string if it has one, else the text it covers ("3'645 Eur" ->
"EUR 3'645").  `{variable|separator}` joins several items of the same variable with the separator
(default: one blank).  Parity of this mini-language with strusAnalyzer is therefore unpinned; what IS
checked against the oracle is everything the engine contributes: which format applies to which
result/item and what its arguments are (MatchBatch.result_format / item_format).

Wire format (include/strus_pattern_amd.h, "result format strings"): item i with a format handle is
followed by item_format[i,1] records that are the arguments of its format string.
"""


class FormatError(ValueError):
    pass


def parse(fmt):
    """-> list of parts: str (literal) or (variable, separator)"""
    parts, lit, i, n = [], [], 0, len(fmt)
    while i < n:
        c = fmt[i]
        if c == "\\" and i + 1 < n:
            lit.append(fmt[i + 1])
            i += 2
        elif c == "{":
            j = fmt.find("}", i + 1)
            if j < 0:
                raise FormatError("missing '}' in result format string: %r" % fmt)
            if lit:
                parts.append("".join(lit))
                lit = []
            body = fmt[i + 1:j]
            var, _, sep = body.partition("|")
            if not var.strip():
                raise FormatError("empty variable reference in result format string: %r" % fmt)
            parts.append((var.strip(), sep if "|" in body else " "))
            i = j + 1
        else:
            lit.append(c)
            i += 1
    if lit:
        parts.append("".join(lit))
    return parts


class Item:
    __slots__ = ("name", "value", "ordpos", "ordend", "origseg", "origpos", "origendseg", "origend")

    def __init__(self, name, value, rec):
        self.name, self.value = name, value
        self.ordpos, self.ordend, self.origseg, self.origpos, self.origendseg, self.origend = (int(x) for x in rec)

    def text(self, src):
        return src[self.origpos:self.origend].decode("utf-8", "replace")

    def __repr__(self):
        return "Item(%s=%r [%d,%d))" % (self.name, self.value, self.origpos, self.origend)


class Result(Item):
    __slots__ = ("items",)

    def __init__(self, name, value, rec, items):
        Item.__init__(self, name, value, rec)
        self.items = items

    def __repr__(self):
        return "Result(%s=%r [%d,%d) %r)" % (self.name, self.value, self.origpos, self.origend, self.items)


class Formatter:
    """Builds PatternMatcherResult-like objects (name, value, positions, items) from the arrays."""

    def __init__(self, pattern_name, variable_name, format_string):
        self.pattern_name, self.variable_name, self.format_string = pattern_name, variable_name, format_string
        self._parsed = {}

    def _parts(self, handle):
        p = self._parsed.get(handle)
        if p is None:
            p = self._parsed[handle] = parse(self.format_string(handle) or "")
        return p

    def evaluate(self, handle, args, src):
        out = []
        for part in self._parts(handle):
            if isinstance(part, str):
                out.append(part)
            else:
                var, sep = part
                out.append(sep.join(a.value if a.value is not None else a.text(src) for a in args if a.name == var))
        return "".join(out)

    def gather(self, items, item_format, begin, end, src):
        """patternMatcher.cpp:164-190 over the flattened records"""
        out, i = [], begin
        while i < end:
            rec = items[i]
            fmt, nsub = (int(item_format[i][0]), int(item_format[i][1])) if item_format is not None else (0, 0)
            value = None
            if fmt:
                value = self.evaluate(fmt, self.gather(items, item_format, i + 1, i + 1 + nsub, src), src)
                i += nsub
            out.append(Item(self.variable_name(int(rec[0])), value, rec[1:7]))
            i += 1
        return out

    def results(self, results, items, result_format, item_format, src):
        """patternMatcher.cpp:248-269"""
        out = []
        for ri, r in enumerate(results):
            ib, ic = int(r[7]), int(r[8])
            lst = self.gather(items, item_format, ib, ib + ic, src)
            fmt = int(result_format[ri]) if result_format is not None else 0
            if fmt:
                out.append(Result(self.pattern_name(int(r[0])), self.evaluate(fmt, lst, src), r[1:7], []))
            else:
                out.append(Result(self.pattern_name(int(r[0])), None, r[1:7], lst))
        return out
