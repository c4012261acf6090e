"""strusPatternMatch-like command line: load a rule program, run documents
through the MI355X lexer + rule automaton, print tokens (-K) and results in the reference tool's
listing format (doc/webpage/introduction_struspattern.htm:178-260).

    python -m struspattern_amd.match -p program.rul [-K] [-o ORIGIN] file [file ...]

Every file is one document (plain UTF-8 text; the reference tool additionally segments XML, which is
outside this engine -- pass -o to add the offset of the text inside its original file to the printed
positions).  All files go to the GPU as one batch.  There is no CPU fallback.
"""
import argparse
import sys

import numpy as np

import struspattern_amd as spa
from struspattern_amd import rulelang


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m struspattern_amd.match", description=__doc__.split("\n\n")[0])
    ap.add_argument("-p", "--program", required=True, help="rule program file")
    ap.add_argument("-K", "--tokens", action="store_true", help="also print the tokens recognized")
    ap.add_argument("-o", "--origin", type=int, default=0, help="offset added to the printed byte positions")
    ap.add_argument("-d", "--device", type=int, default=0)
    ap.add_argument("files", nargs="+")
    args = ap.parse_args(argv)

    with open(args.program, encoding="utf-8") as f:
        program = f.read()
    lx, mt = spa.PatternLexerInstance(), spa.PatternMatcherInstance()
    prg = rulelang.load(program, lx, mt)

    docs = []
    for fn in args.files:
        with open(fn, "rb") as f:
            docs.append(f.read())
    text = b"".join(docs)
    offs = np.zeros(len(docs) + 1, np.uint64)
    offs[1:] = np.cumsum([len(d) for d in docs])

    lb = lx.createContext(args.device).matchDocs(text, offs)
    mb = mt.createContext(args.device).matchDocs(lb.lexems, lb.doc_offsets)
    for di, fn in enumerate(args.files):
        print("%s:" % fn)
        doc = docs[di]
        if args.tokens:
            for line in rulelang.format_tokens(prg, doc, lb.doc(di)):
                print(line)
        res = mb.doc(di)
        for line in rulelang.format_results(doc, res, mb.items, mt.patternName, mt.variableName, origin=args.origin,
                                            result_format=mb.result_format, item_format=mb.item_format, format_string=mt.formatString,
                                            first_result=int(mb.doc_offsets[di])):
            print(line)
    print("OK done")
    return 0


if __name__ == "__main__":
    sys.exit(main())
