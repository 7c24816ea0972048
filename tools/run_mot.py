#!/usr/bin/env python3
"""Run MOT-format sequences through the drop-in BUSCA and score / compare the result files (busca_amd/harness.py).

    python tools/run_mot.py --data-root /data/MOT17/train --seqs MOT17-02-FRCNN MOT17-04-FRCNN --out out/mot17 \\
           --config busca_amd/configs/strongsort_mot17.yml [--weights model_busca.pth] [--tracker pkg.mod:factory] \\
           [--compare-with reference_out/mot17] [--gpus N]
    python tools/run_mot.py --synthetic /tmp/syn --out /tmp/syn_out          # self-test without any dataset

--tracker names a callable `factory(model, seq, args) -> tracker` (object with update(frame_bgr, dets[n,5]) ->
[(id, tlwh, score)]); with the reference's adapters on PYTHONPATH that is a few-line shim over their tracker class.
Default: busca_amd.harness.LiteTracker.  --gpus N: sequences are assigned to ranks by busca_amd.sharding.assign_sequences
(launch with torch.distributed.run; one process per GPU, no collective on the data path).
--compare-with: a folder of result files from a run of the reference; the check is file equality ("HOTA/IDF1 identical").
"""
import argparse
import importlib
import json
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from busca_amd import harness, option, sharding  # noqa: E402
from busca_amd.network import BUSCA  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--data-root")
    ap.add_argument("--seqs", nargs="*")
    ap.add_argument("--synthetic", help="write + run a synthetic MOT-format sequence under this directory")
    ap.add_argument("--out", required=True)
    ap.add_argument("--config", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "busca_amd", "configs", "strongsort_mot17.yml"))
    ap.add_argument("--weights")
    ap.add_argument("--tracker")
    ap.add_argument("--format", choices=["bytetrack", "strongsort"], default="bytetrack")
    ap.add_argument("--compare-with")
    ap.add_argument("--trackeval", action="store_true", help="also score the written files with TrackEval's MOTChallenge2DBox evaluator (HOTA / CLEAR / Identity) "
                                                              "when the package imports - the evaluator GHOST calls (adapters/GHOST/src/eval_track_eval.py:70)")
    ap.add_argument("--assert-identical", action="store_true", help="with --compare-with: exit 1 unless every result file equals the reference run's byte for byte")
    ap.add_argument("--busca-thresh", type=float)
    ap.add_argument("--max-frames", type=int)
    ap.add_argument("--raw-probs", action="store_true", help="select_highest_candidate=False (thresholds the raw probability)")
    a = ap.parse_args()

    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))
    targs, _ = option.load_args_from_config(a.config)
    margs = targs.transformer if hasattr(targs, "transformer") else targs
    margs.device = dev
    if not hasattr(margs, "reid_weights_file"):
        margs.reid_weights_file = "no"
    model = BUSCA(margs).to(dev).eval()
    if a.weights:
        model.load_pretrained(a.weights)
    else:
        print("run_mot: no --weights given: the model runs on seeded random weights (plumbing / parity runs only)", file=sys.stderr)

    if a.synthetic:
        seq_dirs = [harness.write_synthetic_sequence(a.synthetic)]
    else:
        names = a.seqs or sorted(d for d in os.listdir(a.data_root) if os.path.isdir(os.path.join(a.data_root, d)))
        seq_dirs = [os.path.join(a.data_root, n) for n in names]
    seqs = [harness.load_sequence(d) for d in seq_dirs]
    mine = sharding.assign_sequences([len(s) for s in seqs], world)[rank]

    targ = types.SimpleNamespace(seq_len=getattr(targs, "seq_len", 11), num_candidates=getattr(targs, "num_candidates", 5),
                                 use_broader_memory=getattr(targs, "use_broader_memory", True),
                                 select_highest_candidate=False if a.raw_probs else getattr(targs, "select_highest_candidate", True),
                                 busca_thresh=a.busca_thresh if a.busca_thresh is not None else getattr(targs, "busca_thresh", 0.5),
                                 match_thresh=0.8, track_thresh=0.5, det_thresh=0.1, max_time_lost=30)
    factory = None
    if a.tracker:
        mod, fn = a.tracker.split(":")
        factory = getattr(importlib.import_module(mod), fn)
    for i in mine:
        seq = seqs[i]
        tracker = factory(model, seq, targ) if factory else harness.LiteTracker(model, targ)
        n = harness.run_sequence(seq, tracker, os.path.join(a.out, seq.name + ".txt"), fmt=a.format, max_frames=a.max_frames)
        print("rank %d: %s -> %d rows" % (rank, seq.name, n), file=sys.stderr)
    report = {"scores": harness.evaluate(a.out, [seqs[i] for i in mine]), "trackeval_available": harness.trackeval_available()}
    if a.trackeval:
        gt_root = a.synthetic if a.synthetic else a.data_root
        te = harness.run_trackeval(a.out, [seqs[i] for i in mine], gt_root)
        report["trackeval"] = te if te is not None else "TrackEval is not importable here: built-in CLEAR-MOT / IDF1 / HOTA restatements only"
    if a.compare_with:
        report["compare"] = harness.compare_runs(a.out, a.compare_with)
    print(json.dumps(report, indent=1))
    if a.compare_with and a.assert_identical:
        if not report["compare"]["all_identical"]:
            print("run_mot: result files differ from the reference run", file=sys.stderr)
            sys.exit(1)


if __name__ == "__main__":
    main()
