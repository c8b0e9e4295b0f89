"""Stand-in for the handful of torch_geometric symbols the reference model files import."""
