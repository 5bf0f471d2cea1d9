"""hand_tracking_samples_amd -- MI355X-native per-frame hand-tracking hot path (CNN forward + pose solver)."""
