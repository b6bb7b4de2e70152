"""MI355X-native drop-in for the detection hot path of ssds.pytorch (package name kept: ``ssds``)."""
__version__ = "0.1.0"
