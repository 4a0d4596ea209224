"""Merge the winners of a dense sweep (17 columns) and a grouped sweep (19 columns) into the preloaded tiling table
deepgemm_ascend_amd/tuned/mi355x.csv (19 columns; dense rows get groups = 1, contiguous = 0).
Usage: python scripts/build_tuned_table.py <dense_winners.csv> <grouped_winners.csv> [out.csv]"""
import sys
from pathlib import Path

HEAD = ("m,n,k,m1,n1,k1,kernelSerial,paddingTagA,paddingTagB,paddingTagC,blockDim,"
        "splitkFactor,stages,swizzleOffset,wavesM,wavesN,dispatchPolicyTag,groups,contiguous")


def rows(path, pad):
    out = []
    for line in Path(path).read_text().strip().splitlines()[1:]:
        cols = line.split(",")
        out.append(",".join(cols + pad[len(cols) - 17:] if len(cols) < 19 else cols))
    return out


def main():
    dense, grouped = sys.argv[1], sys.argv[2]
    out = Path(sys.argv[3]) if len(sys.argv) > 3 else Path(__file__).resolve().parent.parent / "deepgemm_ascend_amd/tuned/mi355x.csv"
    seen, body = set(), []
    for r in rows(dense, ["1", "0"]) + rows(grouped, []):
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
