"""Loader for the strusPattern rule language (the caller pipeline).

The reference ships no grammar, only the worked example of its web page
(doc/webpage/introduction_struspattern.htm:76-161) -- the program below is restated from that text:

    NAME ^LEVEL : /regex/ | /regex/ ... ;          token (lexem) declarations; the delimiter is the first
    NAME ^LEVEL : @regex@ ;                        non-blank character after ':' or '|'
    [.]Name = op( [var=]arg, ... | range ) ["format"] ;
                                                   op: sequence, sequence_imm, sequence_struct, within,
                                                   within_struct, any, and; arg: a token name, a pattern
                                                   name, TOKEN "symbol", or a nested op(...); a leading
                                                   '.' makes the pattern private (not reported)
    Name = TOKEN ["format"] ;                      a pattern that is one token

`~N` (edit distance) after a regex and result format strings are parsed; edit distance is rejected
(§8(f) row 2, not built), the format string is passed through to definePattern unchanged.

The loader only *drives* the two interfaces (PatternLexerInstance / PatternMatcherInstance method
names), so the same program text configures the MI355X engine and, in the tests, the CPU oracle.
"""
import re

OPS = ("sequence_struct", "sequence_imm", "sequence", "within_struct", "within", "any", "and")


class RuleLangError(ValueError):
    pass


class _Scanner:
    def __init__(self, text):
        self.s = text
        self.i = 0

    def skip(self):
        s, n = self.s, len(self.s)
        while self.i < n:
            c = s[self.i]
            if c in " \t\r\n":
                self.i += 1
            elif c == "#":
                while self.i < n and s[self.i] != "\n":
                    self.i += 1
            else:
                break

    def eof(self):
        self.skip()
        return self.i >= len(self.s)

    def peek(self):
        self.skip()
        return self.s[self.i] if self.i < len(self.s) else ""

    def expect(self, ch):
        if self.peek() != ch:
            self.fail("expected '%s'" % ch)
        self.i += 1

    def accept(self, ch):
        if self.peek() == ch:
            self.i += 1
            return True
        return False

    def ident(self):
        self.skip()
        m = re.compile(r"[A-Za-z_][A-Za-z0-9_]*").match(self.s, self.i)
        if not m:
            self.fail("identifier expected")
        self.i = m.end()
        return m.group(0)

    def number(self):
        self.skip()
        m = re.compile(r"[0-9]+").match(self.s, self.i)
        if not m:
            self.fail("number expected")
        self.i = m.end()
        return int(m.group(0))

    def delimited(self):
        """/regex/ or @regex@ ...: the first non-blank character is the delimiter; no escape of it inside."""
        self.skip()
        d = self.s[self.i]
        if d.isalnum() or d in " \t\r\n;|":
            self.fail("regular expression delimiter expected")
        j = self.s.find(d, self.i + 1)
        if j < 0:
            self.fail("unterminated regular expression")
        body = self.s[self.i + 1:j]
        self.i = j + 1
        return body

    def string(self):
        self.skip()
        q = self.s[self.i]
        if q not in "\"'":
            self.fail("string expected")
        j = self.i + 1
        out = []
        while j < len(self.s) and self.s[j] != q:
            if self.s[j] == "\\" and j + 1 < len(self.s):
                j += 1
            out.append(self.s[j])
            j += 1
        if j >= len(self.s):
            self.fail("unterminated string")
        self.i = j + 1
        return "".join(out)

    def fail(self, msg):
        line = self.s.count("\n", 0, self.i) + 1
        raise RuleLangError("rule program line %d: %s" % (line, msg))


class Program:
    """A parsed rule program bound to a lexer instance and a matcher instance."""

    def __init__(self):
        self.lexem_id = {}        # token name -> lexem id
        self.lexem_name = {}      # lexem id (incl. symbol ids) -> printable name
        self.patterns = []        # pattern names in definition order
        self.symbols = {}         # (token name, text) -> symbol id
        self.variables = []       # variable names in order of first use
        self.formats = []         # non-empty format strings in definePattern order (format handle = index+1)
        self._next_id = 1

    def _new_id(self):
        i = self._next_id
        self._next_id += 1
        return i


def load(text, lexer, matcher, posbind="content"):
    """Parses `text` and defines its tokens on `lexer` and its patterns on `matcher` (both are compiled
    on return).  Returns the Program (name tables)."""
    prg = Program()
    sc = _Scanner(text)
    pending_patterns = []

    while not sc.eof():
        private = sc.accept(".")
        name = sc.ident()
        level = 0
        if sc.accept("^"):
            level = sc.number()
        c = sc.peek()
        if c == ":":
            if private:
                sc.fail("a token cannot be private")
            sc.i += 1
            lid = prg.lexem_id.get(name)
            if lid is None:
                lid = prg._new_id()
                prg.lexem_id[name] = lid
                prg.lexem_name[lid] = name
                if hasattr(lexer, "defineLexemName"):
                    lexer.defineLexemName(lid, name)
            while True:
                expr = sc.delimited()
                if sc.accept("~"):
                    sc.number()
                    sc.fail("edit distance (~N) is not supported")
                lexer.defineLexem(lid, expr, 0, level, posbind)
                if not sc.accept("|"):
                    break
            sc.expect(";")
        elif c == "=":
            sc.i += 1
            pending_patterns.append((name, private, sc.i))
            # skip to the terminating ';' (outside strings); patterns are built after all tokens are known
            depth = 0
            while True:
                ch = sc.peek()
                if ch == "":
                    sc.fail("';' expected")
                if ch in "\"'":
                    sc.string()
                    continue
                sc.i += 1
                if ch == ";" and depth == 0:
                    break
                if ch in "([":
                    depth += 1
                elif ch in ")]":
                    depth -= 1
        else:
            sc.fail("':' or '=' expected after the name")

    def term(tok, symbol=None):
        if tok not in prg.lexem_id:
            return None
        lid = prg.lexem_id[tok]
        if symbol is None:
            return lid
        key = (tok, symbol)
        sid = prg.symbols.get(key)
        if sid is None:
            sid = prg._new_id()
            prg.symbols[key] = sid
            prg.lexem_name[sid] = "%s \"%s\"" % (tok, symbol)
            lexer.defineSymbol(sid, lid, symbol)
        return sid

    pattern_names = set(n for n, _, _ in pending_patterns)

    def expression(sc2):
        """pushes one expression on the matcher's stack"""
        name = sc2.ident()
        if name in OPS and sc2.peek() == "(":
            sc2.i += 1
            argc = 0
            while True:
                save = sc2.i
                var = None
                try:
                    cand = sc2.ident()
                    if sc2.accept("="):
                        var = cand
                    else:
                        sc2.i = save
                except RuleLangError:
                    sc2.i = save
                expression(sc2)
                if var:
                    matcher.attachVariable(var)
                    if var not in prg.variables:
                        prg.variables.append(var)
                argc += 1
                if not sc2.accept(","):
                    break
            rng, card = 0, 0
            if sc2.accept("|"):
                rng = sc2.number()
                if sc2.accept(","):
                    card = sc2.number()
            sc2.expect(")")
            matcher.pushExpression(name, argc, rng, card)
            return
        if sc2.peek() in "\"'":
            sym = sc2.string()
            t = term(name, sym)
            if t is None:
                sc2.fail("symbol of an undefined token '%s'" % name)
            matcher.pushTerm(t)
            return
        t = term(name)
        if t is not None:
            matcher.pushTerm(t)
        elif name in pattern_names:
            matcher.pushPattern(name)
        else:
            sc2.fail("undefined token or pattern '%s'" % name)

    for name, private, pos in pending_patterns:
        sc2 = _Scanner(text)
        sc2.i = pos
        expression(sc2)
        fmt = ""
        if sc2.accept("["):
            fmt = sc2.string()
            sc2.expect("]")
        sc2.expect(";")
        matcher.definePattern(name, fmt, not private)
        if fmt:
            prg.formats.append(fmt)
        if name not in prg.patterns:
            prg.patterns.append(name)

    lexer.compile()
    matcher.compile()
    return prg


def name_tables(prg, matcher):
    """(handle -> pattern name, variable id -> variable name) through the instance's own id lookups."""
    pat = {int(matcher.patternId(n)): n for n in prg.patterns}
    var = {int(matcher.variableId(n)): n for n in prg.variables}
    return (lambda h: pat.get(int(h), "?")), (lambda v: var.get(int(v), "?"))


def formatter(prg, matcher):
    """resultformat.Formatter over the program's name tables and format strings."""
    from . import resultformat
    pn, vn = name_tables(prg, matcher)
    return resultformat.Formatter(pn, vn, lambda h: prg.formats[h - 1] if 0 < h <= len(prg.formats) else None)


def format_tokens(prg, text, lexems, origin=0):
    """The token listing of `strusPatternMatch -K`: "ordpos: NAME text" per lexem."""
    out = []
    for lid, ordpos, origpos, origsize in lexems:
        out.append("%d: %s %s" % (ordpos, prg.lexem_name.get(int(lid), "?"), text[origpos:origpos + origsize].decode("utf-8", "replace")))
    return out


def format_results(text, results, items, pattern_name, variable_name, origin=0, result_format=None, item_format=None, format_string=None, first_result=0):
    """The result listing of strusPatternMatch: "Name [ordpos, origpos]: var [ordpos, origpos, size] 'text' ...".
    With format strings (result_format / item_format of the batch, `results` = rows first_result.. of it) a
    result or item that has a value prints it instead of the text it covers."""
    if result_format is not None:
        from . import resultformat
        fm = resultformat.Formatter(pattern_name, variable_name, format_string)
        out = []
        for r in fm.results(results, items, result_format[first_result:first_result + len(results)], item_format, text):
            parts = ["%s [%d, %d, %d] '%s'" % (i.name, i.ordpos, i.origpos + origin, i.origend - i.origpos, i.value if i.value is not None else i.text(text)) for i in r.items]
            head = "%s [%d, %d]:" % (r.name, r.ordpos, r.origpos + origin)
            if r.value is not None:
                head += " '%s'" % r.value
            out.append((head + " " + " ".join(parts)).rstrip() if parts else head if r.value is not None else head + " ")
        return out
    out = []
    for r in results:
        handle, ordpos, ordend, origseg, origpos, origendseg, origend, ib, ic = (int(x) for x in r)
        parts = []
        for it in items[ib:ib + ic]:
            var, iord, iordend, iseg, ipos, iendseg, iend = (int(x) for x in it)
            parts.append("%s [%d, %d, %d] '%s'" % (variable_name(var), iord, ipos + origin, iend - ipos, text[ipos:iend].decode("utf-8", "replace")))
        out.append("%s [%d, %d]: %s" % (pattern_name(handle), ordpos, origpos + origin, " ".join(parts)))
    return out
