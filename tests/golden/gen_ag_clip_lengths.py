#!/usr/bin/env python3
"""Frames per clip of the Action Genome test split, from the reference's `datasets/AG/ag_test_id.pkl`
(clip id -> list of annotated frames; read in the build container only).  Stored as data
(tests/golden/ag_test_clip_lengths.json): the clip-length distribution drives `tools/ag_split_bench.py`,
the synthetic stand-in for BASELINE.json configs[2]."""
import json
import os
import pickle

HERE = os.path.dirname(os.path.abspath(__file__))
with open("/root/reference/datasets/AG/ag_test_id.pkl", "rb") as f:
    ids = pickle.load(f)
lengths = [len(v) for _, v in sorted(ids.items())]
with open(os.path.join(HERE, "ag_test_clip_lengths.json"), "w") as f:
    json.dump({"source": "datasets/AG/ag_test_id.pkl", "clips": len(lengths), "frames": sum(lengths),
               "frames_per_clip": lengths}, f)
print(len(lengths), sum(lengths), min(lengths), max(lengths))
