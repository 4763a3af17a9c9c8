"""Command-line twins of the reference's dataset scripts (scripts/bvh_to_robot_dataset.py, scripts/smplx_to_robot_dataset.py): same flags, same
folder walk, same output files -- the loops behind them run batched on the GPU.

    python -m gmr_amd.scripts.bvh_to_robot_dataset   --src_folder LAFAN1 --tgt_folder out --robot unitree_g1 [--override]
    python -m gmr_amd.scripts.smplx_to_robot_dataset --src_folder joint_files --tgt_folder out --robot unitree_g1 [--override] [--num_cpus 16]
    python -m gmr_amd.scripts.smoke_test --folder out --robot unitree_g1          # the motion-file checks of scripts/smoke_test.py, headless
"""
