"""datafusion-comet_amd — MI355X-native engine for Comet's scan→filter→aggregate hot path.

The directory name carries a hyphen (it mirrors the reference's name); import it as
``datafusion_comet_amd`` (a two-line shim package at the repo root extends its ``__path__`` here).
"""
