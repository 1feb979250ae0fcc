"""`python -m nimfm_amd train|test ...` -- the reference's end-user commands (`nimfm train`, `nimfm test`,
/root/reference/src/nimfm.nim:72-134) for the solvers that run on the MI355X path (`--solver sgd|adagrad`, plus `mbpsgd` from `nimfm_sparsefm`):
svmlight files are parsed on the GPU (ingest.hip), training runs in libnimfm_hip.so, the test score is
reduced on the device, models are written/read in the reference's text format (`dump`/`load`,
model/factorization_machine.nim:142-220).  Option names follow the reference's proc parameters (cligen
accepts both `--nComponents` and `--n-components`; so does this parser).  The coordinate-descent solvers
(`cd`, `als`) are not on this path: they stay with the reference."""
import argparse
import sys

import numpy as np


def _both(name):
    """--nComponents and --n-components"""
    dashed = "".join("-" + c.lower() if c.isupper() else c for c in name)
    return ["--" + name] if dashed == name else ["--" + name, "--" + dashed]


def _flag(v):
    return str(v).lower() in ("1", "true", "yes", "y", "on")


def _parser():
    ap = argparse.ArgumentParser(prog="nimfm_amd", description="Factorization machines on an MI355X (nimfm's train / test).")
    sub = ap.add_subparsers(dest="cmd", required=True)
    tr = sub.add_parser("train", help="training a factorization machine")
    te = sub.add_parser("test", help="test a factorization machine")
    for p in (tr, te):
        p.add_argument("-t", "--task", required=True, help="r for regression and c for binary classification")
        p.add_argument("--loss", default="squared")
        p.add_argument("--dump", default="")
        p.add_argument("--predict", default="")
        p.add_argument(*_both("nFeatures"), dest="nFeatures", type=int, default=-1)
        p.add_argument("--verbose", type=int, default=1)
    tr.add_argument("--train", required=True)
    tr.add_argument("--test", default="")
    tr.add_argument("--degree", type=int, default=2)
    tr.add_argument(*_both("nComponents"), dest="nComponents", type=int, default=30)
    tr.add_argument("--alpha0", type=float, default=1e-7)
    tr.add_argument("--alpha", type=float, default=1e-5)
    tr.add_argument("--beta", type=float, default=1e-3)
    tr.add_argument(*_both("fitLower"), dest="fitLower", default="explicit")
    tr.add_argument(*_both("fitLinear"), dest="fitLinear", default="true")
    tr.add_argument(*_both("fitIntercept"), dest="fitIntercept", default="true")
    tr.add_argument("--scale", type=float, default=0.1)
    tr.add_argument(*_both("randomState"), dest="randomState", type=int, default=1)
    tr.add_argument("--solver", default="sgd",
                    help="sgd or adagrad; mbpsgd = the mini-batch proximal solver of the reference's nimfm_sparsefm CLI "
                         "(src/nimfm_sparsefm.nim:58-63); cd / als stay with the reference")
    # nimfm_sparsefm train's extra options (src/nimfm_sparsefm.nim:160-170), used by --solver mbpsgd
    tr.add_argument("--gamma", type=float, default=1e-5)
    tr.add_argument("--reg", default="squaredl12", help="l1, l21, squaredl12 or squaredl21")
    tr.add_argument(*_both("miniBatchSize"), dest="miniBatchSize", type=int, default=-1)
    tr.add_argument(*_both("maxIter"), dest="maxIter", type=int, default=100)
    tr.add_argument("--tol", type=float, default=1e-5)
    tr.add_argument("--eta0", type=float, default=0.1)
    tr.add_argument("--scheduling", default="optimal")
    tr.add_argument("--power", type=float, default=1.0)
    tr.add_argument("--threshold", type=float, default=0.1)
    tr.add_argument("--load", default="")
    # this path's own knobs
    tr.add_argument("--mode", default="sequential", choices=["sequential", "minibatch"],
                    help="sequential = the reference's single-thread order; minibatch = the deterministic data-parallel rule")
    tr.add_argument("--batch", type=int, default=8192)
    tr.add_argument(*_both("touchCap"), dest="touchCap", default="1",
                    help="minibatch SGD: steps of a batch on one coordinate summed before averaging sets in (1: the mean; auto: "
                         "about twice the touches per coordinate and batch, nimfm_amd.suggestTouchCap)")
    tr.add_argument(*_both("adaCross"), dest="adaCross", type=float, default=0.0,
                    help="minibatch AdaGrad: weight of the batch's gradient cross products in g_norm (0.1 for large batches)")
    tr.add_argument("--shuffle", default="true")
    te.add_argument("--test", required=True)
    te.add_argument("--load", required=True)
    return ap


def _echo_data_info(X):
    """nimfm.nim:8-13"""
    _, _, data, _ = X.to_host()
    print("   Number of samples  : %d" % X.nSamples)
    print("   Number of features : %d" % X.nFeatures)
    print("   Number of non-zeros: %d" % X.nnz)
    print("   Maximum value      : %s" % (repr(float(data.max())) if len(data) else "-inf"))
    print("   Minimum value      : %s" % (repr(float(data.min())) if len(data) else "inf"))


def _eval(nf, fm, task, test, predict, n_features, verbose):
    """nimfm.nim:16-35; the score is reduced on the device"""
    if verbose > 0:
        print("Load test data")
    X, y = nf.loadSVMLightFile(test, n_features)
    if verbose > 0:
        _echo_data_info(X)
    score = fm.score(X, y)
    print(("Test RMSE: %r" if task == "regression" else "Test Accuracy: %r") % score)
    if predict:
        with open(predict, "w") as f:
            for v in fm.decisionFunction(X):
                f.write(repr(float(v)) + "\n")


def main(argv=None):
    args = _parser().parse_args(argv)
    import nimfm_amd as nf

    task = {"r": "regression", "c": "classification"}.get(args.task, args.task)
    if args.loss not in ("squared", "huber", "squared_hinge", "logistic"):
        raise ValueError("loss %s is not supported" % args.loss)
    if args.cmd == "test":
        fm = nf.load(args.load, False)
        _eval(nf, fm, task, args.test, args.predict, args.nFeatures, args.verbose)
        if args.dump:
            fm.dump(args.dump)
        return 0
    if args.solver not in ("sgd", "adagrad", "mbpsgd"):
        raise ValueError("Solver %s is not supported on this path (sgd, adagrad, mbpsgd; cd / als stay with the reference)" % args.solver)
    if args.load:
        fm = nf.load(args.load, True)
    else:
        fm = nf.newFactorizationMachine(task, degree=args.degree, nComponents=args.nComponents, fitLower=args.fitLower,
                                        fitIntercept=_flag(args.fitIntercept), fitLinear=_flag(args.fitLinear),
                                        warmStart=False, randomState=args.randomState, scale=args.scale)
    X, y = nf.loadSVMLightFile(args.train, args.nFeatures)
    if args.verbose > 0:
        _echo_data_info(X)
    common = dict(maxIter=args.maxIter, eta0=args.eta0, alpha0=args.alpha0, alpha=args.alpha, beta=args.beta,
                  loss=args.loss, verbose=args.verbose, tol=args.tol, shuffle=_flag(args.shuffle), mode=args.mode,
                  batch=args.batch, lossParam=args.threshold)
    if args.solver == "mbpsgd":
        regs = {"l1": nf.newL1, "l21": nf.newL21, "squaredl12": nf.newSquaredL12, "squaredl21": nf.newSquaredL21}
        if args.reg not in regs:
            raise ValueError("reg %s is not supported (l1, l21, squaredl12, squaredl21)" % args.reg)
        opt = nf.newMBPSGD(maxIter=args.maxIter, eta0=args.eta0, alpha0=args.alpha0, alpha=args.alpha, beta=args.beta,
                           gamma=args.gamma, loss=args.loss, reg=regs[args.reg](), miniBatchSize=args.miniBatchSize,
                           scheduling=args.scheduling, power=args.power, verbose=args.verbose, tol=args.tol,
                           shuffle=_flag(args.shuffle), lossParam=args.threshold)
    elif args.solver == "sgd":
        cap = nf.suggestTouchCap(X, args.batch) if str(args.touchCap).lower() == "auto" else float(args.touchCap)
        opt = nf.newSGD(scheduling=args.scheduling, power=args.power, touchCap=cap, **common)
    else:
        opt = nf.newAdaGrad(adaCross=args.adaCross, **common)
    opt.fit(X, y, fm)
    if args.test:
        _eval(nf, fm, task, args.test, args.predict, args.nFeatures, args.verbose)
    if args.dump:
        fm.dump(args.dump)
    return 0


if __name__ == "__main__":
    sys.exit(main())
