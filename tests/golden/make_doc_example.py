"""Refreshes the parts of tests/golden/doc_example.{json,rul} that are taken verbatim from the reference's web
page (doc/webpage/introduction_struspattern.htm:84-97): the URL ^5 expression with the page's whole
alternation of top-level domains, and the CAPWORD / LOWORD expressions with their Unicode property classes.  Run in the build container (the reference tree is not on the GPU box);
the fixtures it writes are committed.  Everything else in the fixture stays as it is (see "adaptations")."""
import html
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
PAGE = "/root/reference/doc/webpage/introduction_struspattern.htm"


def main():
    with open(PAGE, encoding="utf8") as f:
        lines = f.read().split("\n")
    verbatim = {}
    for key in ("URL ^5", "CAPWORD ^1", "LOWORD ^1"):
        verbatim[key] = html.unescape([ln for ln in lines if ln.startswith(key)][0])

    def refreshed(ln):
        for key, text in verbatim.items():
            if ln.startswith(key):
                return text
        return ln
    with open(os.path.join(HERE, "doc_example.json")) as f:
        d = json.load(f)
    prog = d["program"].split("\n")
    prog = [refreshed(ln) for ln in prog]
    d["program"] = "\n".join(prog)
    d["adaptations"] = [a for a in d["adaptations"] if not a.startswith("URL ^5") and not a.startswith("CAPWORD")]
    with open(os.path.join(HERE, "doc_example.json"), "w") as f:
        json.dump(d, f, indent=1, ensure_ascii=False)
        f.write("\n")
    rul = os.path.join(HERE, "doc_example.rul")
    with open(rul) as f:
        text = f.read().split("\n")
    with open(rul, "w") as f:
        f.write("\n".join(refreshed(ln) for ln in text))


if __name__ == "__main__":
    main()
