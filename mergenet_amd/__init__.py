"""mergenet_amd: MI355X-native pixel merger (MergeNet instance-segmentation hot path)."""
