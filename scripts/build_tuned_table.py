"""Merge sweep winners into the preloaded tiling table deepgemm_ascend_amd/tuned/mi355x.csv (19 columns; rows of a 17-column
dense winners file get groups = 1, contiguous = 0).  Files are read in the order given and the FIRST row of a problem
(m, n, k, groups, contiguous) wins -- put the --cold sweep of the short-M shapes in front of the warm dense sweep.
Usage: python scripts/build_tuned_table.py [--out out.csv] winners.csv [winners.csv ...]"""
import sys
from pathlib import Path

HEAD = ("m,n,k,m1,n1,k1,kernelSerial,paddingTagA,paddingTagB,paddingTagC,blockDim,"
        "splitkFactor,stages,swizzleOffset,wavesM,wavesN,dispatchPolicyTag,groups,contiguous")


def rows(path):
    out = []
    for line in Path(path).read_text().strip().splitlines()[1:]:
        cols = line.split(",")
        out.append(",".join(cols + ["1", "0"] if len(cols) == 17 else cols))
    return out


def main():
    args = sys.argv[1:]
    out = Path(__file__).resolve().parent.parent / "deepgemm_ascend_amd/tuned/mi355x.csv"
    if args and args[0] == "--out":
        out, args = Path(args[1]), args[2:]
    seen, body = set(), []
    for f in args:
        for r in rows(f):
            c = r.split(",")
            key = (c[0], c[1], c[2], c[17], c[18])
            if key in seen:
                continue
            seen.add(key)
            body.append(r)
    out.write_text(HEAD + "\n" + "\n".join(body) + "\n")
    print(f"{len(body)} rows -> {out}")


if __name__ == "__main__":
    main()
