"""Import shim: the real package lives in ``datafusion-comet_amd/`` (hyphenated like the reference repo)."""
import os as _os

__path__.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "datafusion-comet_amd"))
