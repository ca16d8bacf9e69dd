"""BUILD-CONTAINER ONLY: import the reference's own arithmetic files from /root/reference through four small
stubs so golden vectors can be produced by reference-authored code. Nothing from the
reference is copied into this repository; only numeric outputs are saved."""
import enum
import importlib.util
import sys
import types

REF = "/root/reference/audiotoken"


def load_reference_modules():
    # (1) StrEnum backport for Python 3.10 (configs.py:2)
    if not hasattr(enum, "StrEnum"):
        class StrEnum(str, enum.Enum):
            def _generate_next_value_(name, start, count, last_values):  # noqa: N805
                return name.lower()

            def __str__(self):
                return str(self.value)
        enum.StrEnum = StrEnum
    # (2) hub downloads executed at class-definition time (configs.py:55-58, 65-70, 114-134)
    import huggingface_hub
    huggingface_hub.hf_hub_download = lambda *a, **k: "/nonexistent/" + k.get("filename", "x")
    huggingface_hub.snapshot_download = lambda *a, **k: "/nonexistent/"
    import transformers  # noqa: F401  (must be imported BEFORE the fake torchaudio: HF probes torchaudio.__spec__)
    import transformers.models.wav2vec2_bert.modeling_wav2vec2_bert  # noqa: F401
    # (3) fake torchaudio (utils.py:7,11)
    if "torchaudio" not in sys.modules:
        ta = types.ModuleType("torchaudio")
        ta.io = types.ModuleType("torchaudio.io")
        ta.io.StreamReader = object
        ta.transforms = types.ModuleType("torchaudio.transforms")
        sys.modules["torchaudio"] = ta
        sys.modules["torchaudio.io"] = ta.io
        sys.modules["torchaudio.transforms"] = ta.transforms
    # (4) synthetic parent package so sub-modules load without running audiotoken/__init__.py
    pkg = types.ModuleType("audiotoken")
    pkg.__path__ = [REF]
    sys.modules["audiotoken"] = pkg
    mods = {}
    for name in ("logger", "configs", "utils", "processors", "modeling_wav2vec2_bert", "datasets"):
        spec = importlib.util.spec_from_file_location(f"audiotoken.{name}", f"{REF}/{name}.py")
        m = importlib.util.module_from_spec(spec)
        sys.modules[f"audiotoken.{name}"] = m
        spec.loader.exec_module(m)
        mods[name] = m
    return mods
