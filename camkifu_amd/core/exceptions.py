"""Error conventions of the finder API (same class names as the reference's core/exceptions.py)."""


class ControllerWarning(Warning):
    pass


class CorrectionWarning(Warning):
    """User corrections that could not be learnt from; carries the (err, exp) move pairs."""

    def __init__(self, corrections, message=None):
        pairs = ", ".join("(err:%s, exp:%s)" % (a, b) for a, b in corrections)
        super().__init__("%s [%s]" % (message if message is not None else "", pairs))
        self.corrections = corrections


class DeletedError(ValueError):
    """A stone was suggested on an intersection the user deleted recently; `locations` holds the
    (r, c) numpy coordinates (or nested DeletedErrors for bulk updates)."""

    def __init__(self, locations, message=None):
        super().__init__(message or "Location deleted by user: locked until its pixels change.")
        self.locations = locations
