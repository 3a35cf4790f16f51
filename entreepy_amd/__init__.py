"""entreepy_amd -- MI355X-native (gfx950) Huffman encode/decode path for typio/entreepy.

The package is a thin host-side mirror of the reference's codec interface
(src/encode.zig:25, src/decode.zig:13) over libentreepy_hip.so, whose C ABI is
declared in include/entreepy_hip.h.  No CPU fallback exists.
"""
from .codec import (  # noqa: F401
    Codebook,
    Context,
    DecodeFlags,
    EmptyInputError,
    EncodeFlags,
    EntreepyError,
    decode,
    default_context,
    encode,
    encode_bound,
    parse_header,
)

__version__ = "0.1.0"
