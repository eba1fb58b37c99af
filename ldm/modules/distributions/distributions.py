"""DiagonalGaussianDistribution (API of ldm/modules/distributions/distributions.py:24-92)."""
import torch


class DiagonalGaussianDistribution(object):
    def __init__(self, parameters, deterministic=False):
        self.parameters = parameters
        self.mean, self.logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.deterministic = deterministic
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)
        if deterministic:
            self.var = self.std = torch.zeros_like(self.mean)

    def sample(self):
        # the reference draws on the CPU generator and copies to the device (:35-37).  Inside a hipGraph capture
        # (the graphed DDIM loop re-samples the hint posterior every denoise step) a host draw + pageable
        # host-to-device copy is illegal -- and would replay ONE frozen noise tensor; there the draw comes from the
        # device generator, whose Philox offset torch advances on every replay: same distribution, fresh noise
        # per step, a different random stream than the reference's (documented in INTEGRATION.md).
        if self.mean.is_cuda and torch.cuda.is_current_stream_capturing():
            return self.mean + self.std * torch.randn(self.mean.shape, device=self.mean.device, dtype=self.mean.dtype)
        return self.mean + self.std * torch.randn(self.mean.shape).to(device=self.parameters.device)

    def mode(self):
        return self.mean
