"""Cross-backend error statistics, restated from the reference's own harness so that the GPU parity tests use
the reference's acceptance bar: Tests::runBackendErrorTest / GpuErrorStats (cpp/tests/testnnevalcanary.cpp:255-373,
thresholds :787-788 fp32 and :806-807 reduced precision) applied to outputs post-processed as NNEvaluator::evaluate
does (cpp/neuralnet/nneval.cpp:960-1254): policy softmax over legal moves, value softmax, score/lead scaling,
ownership tanh. Legality is approximated from the input planes (empty on-board points + pass), which is what
matters for an error statistic."""
import numpy as np

# 99th percentile and max limits: winrate %, lead/score points, top policy %, policy KL
LIMITS_REDUCED = dict(p99=(2.0, 1.00, 2.50, 0.0020), max=(5.0, 3.00, 6.00, 0.0040))  # testnnevalcanary.cpp:806-807
LIMITS_FP32 = dict(p99=(0.45, 0.34, 0.45, 0.0006), max=(1.35, 0.90, 1.35, 0.0012))   # testnnevalcanary.cpp:787-788


def postprocess(out, spatial, info):
    """out: dict of logits from getOutput; spatial [n,S,22]; info: model post-process multipliers."""
    n, S = spatial.shape[0], spatial.shape[1]
    onboard = spatial[:, :, 0] > 0
    empty = onboard & (spatial[:, :, 1] == 0) & (spatial[:, :, 2] == 0)
    legal = np.concatenate([empty, np.ones((n, 1), bool)], axis=1)
    logits = np.where(legal, out["policy"].astype(np.float64), -np.inf)
    logits -= logits.max(axis=1, keepdims=True)
    p = np.exp(logits)
    p /= p.sum(axis=1, keepdims=True)
    v = out["value"].astype(np.float64)
    v = np.exp(v - v.max(axis=1, keepdims=True))
    v /= v.sum(axis=1, keepdims=True)
    sc = out["score"].astype(np.float64)
    softplus = lambda x: np.log1p(np.exp(-np.abs(x))) + np.maximum(x, 0)
    res = dict(policy=p, win=v[:, 0], loss=v[:, 1], noresult=v[:, 2],
               scoreMean=sc[:, 0] * info["scoreMeanMultiplier"], scoreStdev=softplus(sc[:, 1]) * info["scoreStdevMultiplier"],
               lead=sc[:, 2] * info["leadMultiplier"], onboard=onboard)
    if out.get("ownership") is not None:
        res["ownership"] = np.where(onboard, np.tanh(out["ownership"].astype(np.float64)), 0.0)
    return res


def error_stats(base, other):
    wr = np.abs(0.5 * (base["win"] - base["loss"]) - 0.5 * (other["win"] - other["loss"])) + np.abs(base["noresult"] - other["noresult"])
    lead = np.abs(base["lead"] - other["lead"])
    score = np.abs(base["scoreMean"] - other["scoreMean"])
    top = base["policy"].argmax(axis=1)
    idx = np.arange(len(top))
    tpd = np.abs(base["policy"][idx, top] - other["policy"][idx, top])
    pb, po = base["policy"], np.maximum(other["policy"], 1e-300)
    kl = np.where(pb > 1e-30, pb * (np.log(np.maximum(pb, 1e-300)) - np.log(po)), 0.0).sum(axis=1)
    own = np.abs(base["ownership"] - other["ownership"])[base["onboard"]] if "ownership" in base else np.zeros(1)

    def pct(x, q):
        s = np.sort(x)
        return float(s[(len(s) - 1) * q // 100])

    return dict(winrate99=100 * pct(wr, 99), winrateMax=100 * float(wr.max()), lead99=pct(lead, 99), leadMax=float(lead.max()),
                score99=pct(score, 99), scoreMax=float(score.max()), top99=100 * pct(tpd, 99), topMax=100 * float(tpd.max()),
                kl99=pct(kl, 99), klMax=float(kl.max()), own99=pct(own, 99), ownMax=float(own.max()))


def check(stats, limits):
    w99, s99, t99, k99 = limits["p99"]
    wm, sm, tm, km = limits["max"]
    bad = []
    for key, lim in (("winrate99", w99), ("lead99", s99), ("score99", s99), ("top99", t99), ("kl99", k99),
                     ("winrateMax", wm), ("leadMax", sm), ("scoreMax", sm), ("topMax", tm), ("klMax", km)):
        if not stats[key] <= lim:
            bad.append("%s=%.4g > %.4g" % (key, stats[key], lim))
    return bad
