"""Refreshes the parts of tests/golden/doc_example.{json,rul} that are taken verbatim from the reference's web
page (doc/webpage/introduction_struspattern.htm:84-97): the URL ^5 expression with the page's whole
alternation of top-level domains.  Run in the build container (the reference tree is not on the GPU box);
the fixtures it writes are committed.  Everything else in the fixture stays as it is (see "adaptations")."""
import html
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
PAGE = "/root/reference/doc/webpage/introduction_struspattern.htm"


def main():
    with open(PAGE, encoding="utf8") as f:
        lines = f.read().split("\n")
    url5 = html.unescape([ln for ln in lines if ln.startswith("URL ^5")][0])
    with open(os.path.join(HERE, "doc_example.json")) as f:
        d = json.load(f)
    prog = d["program"].split("\n")
    prog = [url5 if ln.startswith("URL ^5") else ln for ln in prog]
    d["program"] = "\n".join(prog)
    d["adaptations"] = [a for a in d["adaptations"] if not a.startswith("URL ^5")]
    with open(os.path.join(HERE, "doc_example.json"), "w") as f:
        json.dump(d, f, indent=1, ensure_ascii=False)
        f.write("\n")
    rul = os.path.join(HERE, "doc_example.rul")
    with open(rul) as f:
        text = f.read().split("\n")
    with open(rul, "w") as f:
        f.write("\n".join(url5 if ln.startswith("URL ^5") else ln for ln in text))


if __name__ == "__main__":
    main()
