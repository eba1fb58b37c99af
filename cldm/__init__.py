"""Drop-in mirror of the reference's `cldm` package for the CtrLoRA hot path: same import paths,
class names, constructor kwargs and state-dict keys; execution on the MI355X HIP engine."""
