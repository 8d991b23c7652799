#!/usr/bin/env python3
"""The reference's main.py flow (main.py:9-93) without its prompts, on the drop-in Recommender:
read dataset.csv / users.csv / queries.csv / utility_matrix.csv from a directory (the layout
resources/generator.py writes), predict every missing rating on the GPU, print a summary and
optionally write final_predictions.csv.   usage: run_csv.py DIR [--perm 180] [--seed S] [--out FILE]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "query-recommendation-system_amd"))
import numpy as np  # noqa: E402
import pandas as pd  # noqa: E402
import recommender as R  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--perm", type=int, default=180)
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    for name in ("dataset", "users", "queries", "utility_matrix"):
        path = os.path.join(a.dir, name + ".csv")
        if not os.path.exists(path):
            sys.exit("Error: {} doesn't exist!".format(path))       # main.py:19-21 exits with 1
    rec = R.Recommender()
    dataset = pd.read_csv(os.path.join(a.dir, "dataset.csv"))
    if dataset.shape[0] == 0:
        sys.exit(2)                                                 # main.py:27-30
    rec.datasetFeatures = list(dataset.columns)[1:]
    users = pd.read_csv(os.path.join(a.dir, "users.csv"), header=None)
    queries, qids = rec.parse_queries(os.path.join(a.dir, "queries.csv"))
    ratings = pd.read_csv(os.path.join(a.dir, "utility_matrix.csv"))
    ratings.insert(0, "user", users[0].to_numpy())
    ratings.columns = ["user"] + qids
    rec.init(users, queries, qids, dataset, ratings)
    R.PERM = a.perm
    if a.seed is not None:
        np.random.seed(a.seed)
    t0 = time.time()
    to_predict, predictions, missed = rec.compute_scores()
    print("\nFINAL PREDICTIONS [{} scores to predict, {} scores missed - {}% miss] in {:.2f}s:".format(
        len(to_predict), len(missed), round(len(missed) / max(len(to_predict), 1) * 100, 3), time.time() - t0))
    print(predictions)
    if a.out:
        predictions.to_csv(a.out)
        print("Final utility matrix saved in", a.out)


if __name__ == "__main__":
    main()
