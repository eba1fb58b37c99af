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
        # the reference draws on the CPU generator and copies to the device (:35-37)
        return self.mean + self.std * torch.randn(self.mean.shape).to(device=self.parameters.device)

    def mode(self):
        return self.mean
