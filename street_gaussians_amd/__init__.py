"""MI355X-native differentiable Gaussian rasterizer (drop-in for street_gaussians' hot path)."""
