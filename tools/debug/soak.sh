# two short end-to-end trainings at the BASELINE batch (32 + 128) with everything on (autotuner / tile database, Winograd, folded finalizers, eval-mode epilogue)
for net in effinetv2 repvgg; do
  python train.py --config settings/300WLP_AFLW2000.yml --network $net --synthetic --exp_root /tmp/soak_$net --batch_size 32 --ulb_batch_ratio 4 \
    --stage1_iteration 10 --max_iteration 60 --log_frequency 10 --val_frequency 10 --save_frequency 50 --dynamic_thres --cutout_aug --cutmix_aug --rotate_aug --left_ratio 0.95 \
    > gpurun_out/soak_$net.log 2>&1
  echo "$net rc=$?"; grep -i "nan\|inf \|error\|traceback" gpurun_out/soak_$net.log | head -5; tail -4 gpurun_out/soak_$net.log
done
