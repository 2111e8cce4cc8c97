"""MI355X-native per-realization hot path of ParELAGMC (SPDE Matérn sampler + mixed Darcy QoI)."""
__version__ = "0.1.0"
