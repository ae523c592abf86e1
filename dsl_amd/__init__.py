"""dsl_amd: MI355X-native FCOS R50-FPN teacher-student training step (hot path of chenbinghui1/DSL)."""
__version__ = '0.1.0'
