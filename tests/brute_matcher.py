"""TEST INFRASTRUCTURE ONLY -- the reference's INDEPENDENT expression-tree matcher, restated.

tests/randomExpressionTreeMatch/src/testRandomExpressionTreeMatch.cpp of the reference checks the
rule automaton against a second, structurally different implementation of the operator semantics: a
recursive matcher that walks the document for every candidate tree (`matchTree`, :573-742) driven by a
key-token index (`fillKeyTokens` :366-423, `processDocumentAlt` :745-775), with trees generated from
the documents' own content so that they can match (`createRandomTree` :239-338).  This module restates
those functions one to one (same control flow, same tie breaks) so that the CPU oracle -- a restatement
of ruleMatcherAutomaton.cpp -- can be checked against code of the reference that does NOT share its
algorithm.  The reference compares the two as SETS of strings `name_ordpos..ordend(seg|ofs .. seg|ofs)`
(:871-900) and documents that they disagree for `within` with non-disjoint arguments
(doc/webpage/introduction_struspattern.htm:366-375).

Nothing here is imported by the product."""
import itertools

import numpy as np

DELIM = 1 << 24  # termId(SentenceDelim, 0), tests/utils/testUtils.cpp:74-77

OPS = ("sequence", "sequence_struct", "within", "within_struct", "any")  # randomOp(), :207-219 (no `and`, no `sequence_imm`)


class Tree:
    """TreeNode (:50-193): a term leaf, or an operator node with arguments, range and cardinality."""

    def __init__(self, term=0, op=None, args=(), range_=0, cardinality=0):
        self.term = term
        self.op = op
        self.args = list(args)
        self.range = range_
        self.cardinality = cardinality
        self.variable = None
        self.name = None

    def is_term(self):
        return bool(self.term) and not self.args


class Rand:
    """RANDINT(MIN,MAX) = rand() % (MAX-MIN) + MIN (:48) and the Zipf draws of GlobalContext (:193-236),
    on a seeded generator (the reference seeds rand() from the calendar date)."""

    def __init__(self, seed, nof_features):
        from struspattern_amd import synth
        self.rng = np.random.default_rng(seed)
        self._zs = synth.zipf_sample
        self.featcum = synth.zipf_cum(nof_features, 0.8)
        self.rangecum = synth.zipf_cum(20, 1.3)
        self.selopcum = synth.zipf_cum(5)
        self.argccum = synth.zipf_cum(5, 1.3)

    def randint(self, lo, hi):
        return int(self.rng.integers(lo, hi)) if hi > lo else lo

    def random_range(self):
        return int(self._zs(self.rangecum, self.rng)) - 1

    def random_op(self):
        return OPS[int(self._zs(self.selopcum, self.rng)) - 1]

    def random_argc(self):
        rt = int(self._zs(self.argccum, self.rng))
        if rt == 1 and self.randint(1, 5) >= 2:
            rt += 1
        return rt


def create_random_tree(rnd, doc, docitr, depth=0, maxdepth=3):
    """createRandomTree (:239-338).  doc: list of (termid, pos); docitr: one-element list (by reference).
    The reference's `RANDINT(1,5-depth)` ends every tree at depth 3; `maxdepth` lifts that bound the same way."""
    if rnd.randint(1, maxdepth + 2 - depth) == 1:
        if docitr[0] >= len(doc):
            return None
        rt = Tree(term=doc[docitr[0]][0])
    else:
        argc = rnd.random_argc()
        range_ = argc + rnd.random_range()
        op = rnd.random_op()
        args = []
        rangesum = 0
        ai = 0
        while docitr[0] < len(doc) and ai < argc:
            arg = create_random_tree(rnd, doc, docitr, depth + 1, maxdepth)
            if arg is None:
                return None
            args.append(arg)
            rangesum += arg.range
            ai += 1
            docitr[0] += 1
        if argc == 1 and op in ("sequence_struct", "within_struct") and args and args[0].term == DELIM:
            arg = create_random_tree(rnd, doc, docitr, depth + 1, maxdepth)
            if arg is None:
                return None
            args.append(arg)
            rangesum += arg.range
        range_ += rangesum
        if ai < argc:
            return None
        if op == "sequence_struct":
            args.insert(0, Tree(term=DELIM))
        elif op in ("within", "within_struct"):
            for _ in range(3):
                r1, r2 = rnd.randint(0, argc), rnd.randint(0, argc)
                if r1 != r2:
                    args[r1], args[r2] = args[r2], args[r1]
            if op == "within_struct":
                args.insert(0, Tree(term=DELIM))
        rt = Tree(op=op, args=args, range_=range_, cardinality=0)
    if rnd.randint(1, 10) == 1:
        rt.variable = "v%d" % rnd.randint(1, 10)
    return rt


def create_random_trees(rnd, docs, nof_rules, maxdepth=3, accept=None):
    """createRandomTrees (:340-364): `any` at the top is skipped, the top node carries no variable.
    accept(tree) -> bool filters further (e.g. pairwise disjoint arguments)."""
    out = []
    tries = 0
    while len(out) < nof_rules:
        tries += 1
        if tries > 200 * nof_rules + 1000:
            raise RuntimeError("cannot generate %d trees" % nof_rules)
        doc = docs[(len(out) + tries) % len(docs)]
        tree = create_random_tree(rnd, doc, [0], 0, maxdepth)
        if tree is None or tree.op == "any":
            continue
        if accept is not None and not accept(tree):
            continue
        tree.variable = None
        tree.name = "pattern_%d" % (len(out) + 1)
        out.append(tree)
    return out


def fill_key_tokens(keymap, tree, idx):
    """fillKeyTokens (:366-423): term -> trees it can start."""
    if tree.is_term():
        keymap.setdefault(tree.term, []).append(idx)
    elif tree.op in ("sequence", "sequence_imm"):
        fill_key_tokens(keymap, tree.args[0], idx)
    elif tree.op == "sequence_struct":
        fill_key_tokens(keymap, tree.args[1], idx)
    elif tree.op in ("within", "any"):
        for a in tree.args:
            fill_key_tokens(keymap, a, idx)
    elif tree.op == "within_struct":
        for a in tree.args[1:]:
            fill_key_tokens(keymap, a, idx)
    else:
        raise ValueError("operator %r not implemented by the reference's checker" % tree.op)


def apply_tree(m, tree):
    """createExpression (:425-445) on any object with the PatternMatcherInstanceInterface method names."""
    if tree.is_term():
        m.pushTerm(tree.term)
    else:
        for a in tree.args:
            apply_tree(m, a)
        m.pushExpression(tree.op, len(tree.args), tree.range, tree.cardinality)
    if tree.variable:
        m.attachVariable(tree.variable)


def apply_trees(m, trees, compile_=True):
    """createRules (:447-456)."""
    for t in trees:
        apply_tree(m, t)
        m.definePattern(t.name, "", True)
    if compile_:
        m.compile()


class Match:
    """TreeMatchResult (:514-571)."""
    __slots__ = ("valid", "startidx", "endidx", "ordpos", "ordsize", "items")

    def __init__(self, startidx=0, endidx=0, ordpos=0, ordsize=0, valid=False):
        self.valid = valid
        self.startidx, self.endidx, self.ordpos, self.ordsize = startidx, endidx, ordpos, ordsize
        self.items = []

    def copy(self):
        m = Match(self.startidx, self.endidx, self.ordpos, self.ordsize, self.valid)
        m.items = list(self.items)
        return m

    def join(self, o):
        if not self.valid:
            c = o.copy()
            self.valid, self.startidx, self.endidx, self.ordpos, self.ordsize, self.items = c.valid, c.startidx, c.endidx, c.ordpos, c.ordsize, c.items
        elif o.valid:
            ordend = max(self.ordpos + self.ordsize, o.ordpos + o.ordsize)
            self.ordpos = min(self.ordpos, o.ordpos)
            self.startidx = min(self.startidx, o.startidx)
            self.endidx = max(self.endidx, o.endidx)
            self.ordsize = ordend - self.ordpos
            self.items += o.items


def match_tree(tree, doc, didx, endpos, first_term):
    """matchTree (:573-742).  doc: list of (termid, pos)."""
    rt = Match()
    n = len(doc)
    if tree.is_term():
        if first_term:
            if didx >= n:
                return Match()
            if doc[didx][1] < endpos:
                endpos = doc[didx][1]
        while didx < n:
            if doc[didx][1] > endpos:
                break
            if doc[didx][0] == tree.term:
                if first_term and doc[didx][0] != first_term:
                    didx += 1
                    continue
                rt = Match(didx, didx + 1, doc[didx][1], 1, True)
                break
            didx += 1
    elif tree.op in ("sequence", "sequence_struct"):
        args = tree.args[1:] if tree.op == "sequence_struct" else tree.args
        for a in args:
            ar = match_tree(a, doc, didx, endpos, first_term)
            if not ar.valid:
                rt = Match()
                break
            if ar.ordpos + tree.range < endpos:
                endpos = ar.ordpos + tree.range
            first_term = 0
            rt.join(ar)
            didx = rt.endidx
            nextpos = ar.ordpos + ar.ordsize
            while didx < n and doc[didx][1] < nextpos:
                didx += 1
        if tree.op == "sequence_struct" and rt.valid:
            delim = match_tree(tree.args[0], doc, rt.startidx, rt.ordpos + rt.ordsize, 0)
            if delim.valid and delim.endidx < rt.endidx:
                return Match()
    elif tree.op in ("within", "within_struct"):
        aidx = 0 if tree.op == "within" else 1
        seqop = "sequence" if tree.op == "within" else "sequence_struct"
        # getIndexPermurations (:493-512): every order of the arguments; the candidate that ends first wins, the
        # first one found among equals
        for perm in _index_permutations(aidx, len(tree.args)):
            pargs = ([tree.args[0]] if tree.op == "within_struct" else []) + [tree.args[i] for i in perm]
            alt = Tree(op=seqop, args=pargs, range_=tree.range, cardinality=tree.cardinality)
            cand = match_tree(alt, doc, didx, endpos, first_term)
            if cand.valid and (not rt.valid or cand.endidx < rt.endidx):
                rt = cand
    elif tree.op == "any":
        selected = Match()
        for a in tree.args:
            if rt.valid:
                break
            cand = match_tree(a, doc, didx, endpos, first_term)
            if cand.valid:
                if cand.ordpos + tree.range < endpos:
                    endpos = cand.ordpos + tree.range
                if not selected.valid or cand.endidx < selected.endidx:
                    selected = cand
        rt.join(selected)
    else:
        raise ValueError("operator %r not implemented by the reference's checker" % tree.op)
    if rt.valid and tree.variable:
        rt.items.append((tree.variable, rt.ordpos, rt.ordpos + rt.ordsize, rt.startidx, rt.endidx))
    return rt


def _index_permutations(begin, end):
    """getIndexPermurations (:493-512), same enumeration order."""
    if begin + 1 == end:
        return [[begin]]
    out = []
    for p in _index_permutations(begin + 1, end):
        for t in range(len(p) + 1):
            q = list(p)
            q.insert(t, begin)
            out.append(q)
    return out


def process_document_alt(keymap, trees, doc):
    """processDocumentAlt (:745-775): results as (name, ordpos, ordend, startidx, endidx)."""
    out = []
    for didx, (termid, pos) in enumerate(doc):
        prev = None
        for tidx in keymap.get(termid, ()):
            if prev == tidx:
                continue            # duplicates of redundant key tokens of one rule
            prev = tidx
            t = trees[tidx]
            m = match_tree(t, doc, didx, pos + t.range, termid)
            if m.valid:
                out.append((t.name, m.ordpos, m.ordpos + m.ordsize, m.startidx, m.endidx))
    return out


def result_strings(results):
    """the string form the reference compares as a set (:871-900); origseg is 0 in its documents"""
    return set("%s_%d..%d(0|%d .. 0|%d)" % r for r in results)


def term_sets_disjoint(tree):
    """every operator node's arguments use pairwise disjoint term sets (the case the reference's documentation
    claims agreement for, doc/webpage/introduction_struspattern.htm:366-375)"""
    def terms(t):
        if t.is_term():
            return {t.term}
        s = set()
        for a in t.args:
            s |= terms(a)
        return s

    def ok(t):
        if t.is_term():
            return True
        seen = set()
        for a in t.args:
            ts = terms(a)
            if seen & ts:
                return False
            seen |= ts
        return all(ok(a) for a in t.args)
    return ok(tree)


def delimiter_only_structural(tree):
    """the sentence delimiter appears only as the structure element of *_struct nodes.  As an ordinary argument it
    shares its ordinal position with the token before it, and the automaton processes such pairs in arrival order
    (documented anomaly, doc/webpage/introduction_struspattern.htm:380-392): the two implementations differ there."""
    if tree.is_term():
        return tree.term != DELIM
    args = tree.args[1:] if tree.op in ("sequence_struct", "within_struct") else tree.args
    return all(delimiter_only_structural(a) for a in args)
