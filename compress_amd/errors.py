"""Error classes of the reference (internal/errors/errors.go:36-101, api.go:16-27)."""

Unknown, Internal, Invalid, Deprecated, Corrupted, Closed = range(6)      # errors.go:36-56

_CODE_MAP = {Unknown: "unknown error", Internal: "internal error", Invalid: "invalid argument",
             Deprecated: "deprecated format", Corrupted: "corrupted input", Closed: "closed handler"}


class Error(Exception):
    """errors.Error{Code, Pkg, Msg} (errors.go:67-88); satisfies compress.Error (api.go:16-27)."""

    def __init__(self, code, pkg="", msg=""):
        self.Code, self.Pkg, self.Msg = code, pkg, msg
        Exception.__init__(self, ": ".join(s for s in (pkg, _CODE_MAP.get(code, ""), msg) if s))

    def IsInternal(self): return self.Code == Internal
    def IsInvalid(self): return self.Code == Invalid
    def IsDeprecated(self): return self.Code == Deprecated
    def IsCorrupted(self): return self.Code == Corrupted
    def IsClosed(self): return self.Code == Closed


class EOFError_(Exception):
    """io.EOF"""


class UnexpectedEOF(Exception):
    """io.ErrUnexpectedEOF"""


class DeviceError(RuntimeError):
    """Transport failure below the C ABI (HIP error, missing device, caller I/O error)."""


class LimitError(DeviceError):
    """A chunk (or a plain DEFLATE stream) is larger than the device path decodes in one piece: a limit of this
    library (include/xflate_amd.h, XFG_MAX_CHUNK_*), not an error of the input; the reference has no such limit."""


EOF = EOFError_


def IsInternal(err): return isinstance(err, Error) and err.Code == Internal
def IsInvalid(err): return isinstance(err, Error) and err.Code == Invalid
def IsDeprecated(err): return isinstance(err, Error) and err.Code == Deprecated
def IsCorrupted(err): return isinstance(err, Error) and err.Code == Corrupted
def IsClosed(err): return isinstance(err, Error) and err.Code == Closed
