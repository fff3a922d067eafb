"""isaac_rover_orbit_amd -- MI355X-native AAURoverEnv-v0 hot path."""
__version__ = "0.5.0"
