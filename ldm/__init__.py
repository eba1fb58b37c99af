"""Import-path mirror of the reference's `ldm` package for the CtrLoRA hot path.

Only what cldm.* needs is provided; module trees keep the reference's parameter names so its
checkpoints load unchanged, execution goes to the HIP engine (ctrlora_amd.engine).
"""
