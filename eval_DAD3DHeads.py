#!/usr/bin/env python
"""eval_DAD3DHeads.py — same command line and output file as the reference's script (DAD-3DHeads benchmark submission JSON of
predicted rotation matrices for test/test.json), with the crops batched through the MI355X hot path:

    python eval_DAD3DHeads.py <exp_detail>/<date>/best --is_full_range --config settings/DAD3DHeads_WildHead.yml --network effinetv2
"""
import json
import os

from semiuhpe_amd.src.agent import get_agent
from semiuhpe_amd.src.config import get_config
from semiuhpe_amd.src.dad3d_eval import predict_rotations, write_submission


def main():
    config = get_config("test")
    agent = get_agent(config)
    agent.load_ckpt(config.ckpt)
    db_path = config.data_dir_DAD3DHeads
    with open(os.path.join(db_path, "test", "test.json")) as fh:
        items = json.load(fh)
    path = write_submission(predict_rotations(agent, items, db_path), config)
    print(f"wrote {path} ({len(items)} items)")


if __name__ == "__main__":
    main()
