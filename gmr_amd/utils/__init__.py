"""The import paths of general_motion_retargeting.utils (lafan1, smpl) with the reference's own function signatures and return shapes, so that a
script keeps its loops and only swaps the package name.  The batched forms live in gmr_amd.bvh / gmr_amd.smplx_adapter."""
