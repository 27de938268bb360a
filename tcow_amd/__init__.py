"""tcow_amd: MI355X-native (gfx950) implementation of the TCOW Seeker / QueryMaskTracker hot path."""
__version__ = '0.1.0'
