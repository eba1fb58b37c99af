#!/bin/bash
# f4: the training / sampling scripts end to end on the GPU box, on synthetic checkpoints and data
mkdir -p gpurun_out/r02_scripts
O=$GRAFT_REPO_ROOT/gpurun_out/r02_scripts
export CTRLORA_SYNTHETIC_TOKENIZER=1
cd /tmp && export TMPDIR=/tmp
S=/tmp/ctrlora_synth
(cd $GRAFT_REPO_ROOT && timeout 600 python tests/tools/make_synthetic_assets.py --out $S --n 8) > $O/assets.log 2>&1; tail -1 $O/assets.log | cut -c1-300
mkdir -p /tmp/work && cd /tmp/work
R=$GRAFT_REPO_ROOT
timeout 900 python $R/scripts/train_ctrlora_finetune.py --dataroot $S/custom --config $S/finetune_narrow.yaml --sd_ckpt $S/sd_synth.ckpt --cn_ckpt $S/basecn_synth.ckpt --bs 2 --max_steps 4 --precision 16 --ckpt_logger_freq 4 --img_logger_freq 4 --lr 1e-4 -n f4_ft > $O/train_finetune.log 2>&1; echo "finetune rc=$?" | tee -a $O/train_finetune.log; tail -4 $O/train_finetune.log
CK=$(find runs/f4_ft -name "*.ckpt" | head -1); echo "ckpt: $CK" | tee $O/ckpt.txt
find runs/f4_ft -name "*.png" | head -20 >> $O/ckpt.txt
timeout 900 python $R/scripts/sample.py --dataroot $S/custom --config $S/finetune_narrow.yaml --ckpt "$CK" --n_samples 2 --save_dir /tmp/work/samples --ddim_steps 10 > $O/sample.log 2>&1; echo "sample rc=$?" | tee -a $O/sample.log; ls samples/sample samples/control 2>/dev/null | tee -a $O/sample.log | tail -6
timeout 900 python $R/scripts/train_ctrlora_pretrain.py --dataroot $S/multigen --config $S/pretrain_narrow.yaml --sd_ckpt $S/sd_synth.ckpt --cn_ckpt $S/basecn_synth.ckpt --bs 2 --max_steps 4 --precision 16 --ckpt_logger_freq 100 --img_logger_freq 100 --num_workers 2 --lr 1e-4 -n f4_pt > $O/train_pretrain.log 2>&1; echo "pretrain rc=$?" | tee -a $O/train_pretrain.log; tail -4 $O/train_pretrain.log
timeout 900 python $R/scripts/train_ctrlora_finetune.py --dataroot $S/multigen --multigen20m --task canny --config $S/finetune_full_narrow.yaml --sd_ckpt $S/sd_synth.ckpt --cn_ckpt $S/basecn_synth.ckpt --bs 2 --max_steps 3 --precision 32 --ckpt_logger_freq 100 --img_logger_freq 100 --lr 1e-4 -n f4_full > $O/train_full.log 2>&1; echo "full-finetune rc=$?" | tee -a $O/train_full.log; tail -4 $O/train_full.log
cp samples/sample/0.png $O/sample_0.png 2>/dev/null; true
