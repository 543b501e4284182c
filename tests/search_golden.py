"""Parser and comparison for the output of the reference's `runsearchtestsv8` (cpp/tests/testsearchv8.cpp; setup
cpp/tests/testsearchcommon.cpp:193-248): a sequence of search reports, each with the root line and one line per child
(`LOC : T ..c W ..c S ..c (...) LCB ..c P ..% WF .. PSV .. N  <visits> -- <pv>`). A search amplifies last-digit
differences of the net, so visit counts are compared statistically, against the spread the reference itself shows
between its own fp32 and fp16 goldens (cpp/tests/results/runSearchTestsV8Bin.txt vs runSearchTestsV8FP16.txt)."""
import re

_ROOT = re.compile(r"^: T\s+(-?[\d.]+)c W\s+(-?[\d.]+)c S\s+(-?[\d.]+)c \(.*?\) N\s+(\d+)\s+--\s*(.*)$")
_CHILD = re.compile(r"^([A-T]\d+|pass)\s*: T\s+(-?[\d.]+)c W\s+(-?[\d.]+)c .*? LCB\s+(-?[\d.]+)c P\s+([\d.]+)% .*? N\s+(\d+)\s+--")


def parse(text):
    """-> list of searches: dict(root_T, root_N, pv, children=[(loc, T, prior%, N, LCB), ...]) in file order. The reference prints the
    children by PLAY SELECTION VALUE (search/searchresults.cpp: visits, with the best lower confidence bound among the well-visited
    moves promoted to the top), so children[0] - "the best move" - need not be the most visited child."""
    out, cur = [], None
    for line in text.replace("\r", "\n").splitlines():
        m = _ROOT.match(line)
        if m:
            cur = dict(root_T=float(m.group(1)), root_N=int(m.group(4)), pv=m.group(5).split(), children=[])
            out.append(cur)
            continue
        m = _CHILD.match(line)
        if m and cur is not None:
            cur["children"].append((m.group(1), float(m.group(2)), float(m.group(5)), int(m.group(6)), float(m.group(4))))
    return [s for s in out if s["children"]]


def compare(a, b):
    """Statistics of search list `a` against `b` (same positions, same seeds): fraction with the same best move, the
    largest / mean absolute difference of the best move's visit share, the largest root-utility difference (in c, i.e.
    hundredths), the mean total-variation distance of the child visit distributions; `flips` lists every search whose best move
    differs, with both candidates' visit counts in both searches (is it a near-tie, or a different opinion?)."""
    n = min(len(a), len(b))
    same, share, util, tv, flips = 0, [], [], [], []
    for idx, (x, y) in enumerate(zip(a[:n], b[:n])):
        if x["root_N"] != y["root_N"] or x["root_N"] <= 1:
            continue
        bx, by = x["children"][0], y["children"][0]
        same += bx[0] == by[0]
        vy = {c[0]: c[3] for c in y["children"]}
        vx = {c[0]: c[3] for c in x["children"]}
        tot = max(sum(vx.values()), 1), max(sum(vy.values()), 1)
        if bx[0] != by[0]:
            # a flipped best move: how far apart the two candidates are in EACH search - in visits, as a share of that search's child
            # visits (gap_b = b's margin of its own best move over a's best move inside b; gap_a = a's margin the other way inside a), and
            # in the lower confidence bound that decides the order between well-visited moves (lcb_gap_*, in c = hundredths of utility;
            # None when a search did not list the other's move)
            lx = {c[0]: c[4] for c in x["children"]}
            ly = {c[0]: c[4] for c in y["children"]}
            flips.append(dict(index=idx, root_N=x["root_N"], a_best=bx[0], b_best=by[0],
                              lcb_gap_a=abs(lx[bx[0]] - lx[by[0]]) if by[0] in lx else None,
                              lcb_gap_b=abs(ly[by[0]] - ly[bx[0]]) if bx[0] in ly else None,
                              a_visits=(vx.get(bx[0], 0), vx.get(by[0], 0)), b_visits=(vy.get(bx[0], 0), vy.get(by[0], 0)),
                              a_top=[(c[0], c[3]) for c in x["children"][:3]], b_top=[(c[0], c[3]) for c in y["children"][:3]],
                              gap_a=(vx.get(bx[0], 0) - vx.get(by[0], 0)) / tot[0], gap_b=(vy.get(by[0], 0) - vy.get(bx[0], 0)) / tot[1]))
        share.append(abs(vx.get(by[0], 0) / tot[0] - by[3] / tot[1]))
        util.append(abs(x["root_T"] - y["root_T"]))
        keys = set(vx) | set(vy)
        tv.append(0.5 * sum(abs(vx.get(k, 0) / tot[0] - vy.get(k, 0) / tot[1]) for k in keys))
    m = len(share)
    return dict(searches=m, same_best=same / max(m, 1), best_share_max=max(share), best_share_mean=sum(share) / m,
                root_util_max=max(util), root_util_mean=sum(util) / m, tv_mean=sum(tv) / m, tv_max=max(tv), flips=flips)
