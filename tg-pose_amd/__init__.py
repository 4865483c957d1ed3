"""tg-pose_amd: MI355X-native point-cloud forward path of TG-Pose."""
__version__ = "0.1.0"
