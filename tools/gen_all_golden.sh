#!/bin/bash
# Regenerate EVERY fixture under tests/golden/ by executing the reference's own code (build container only: needs
# /root/reference; nothing here travels to the GPU box except the resulting data files).
#   bash tools/gen_all_golden.sh
set -euo pipefail
cd "$(dirname "$0")/.."
export PYTHONPATH="$PWD:$PWD/oracle:$PWD/tools${PYTHONPATH:+:$PYTHONPATH}"
python tools/gen_golden.py                    # backbone_{small,tiny224,h4}, shift_{tiny224,mid320,cfg2}, swin_w*
python tools/gen_golden_swin_net.py           # swin_net_pad120
python tools/gen_golden_swin_net.py --det     # swin_det_r98x118 (the mmdet-style SwinTransformer class)
python tools/gen_golden_annotations.py
python tools/gen_golden_checkpoint.py
python tools/gen_golden_consumers.py
python tools/gen_golden_head_losses.py
python tools/gen_golden_hungarian.py
python tools/gen_golden_mae_heads.py
python tools/gen_golden_mmdet_pure.py
python tools/gen_golden_point_loss.py
python tools/gen_golden_sampler.py
ls -la tests/golden
