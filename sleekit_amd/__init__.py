"""sleekit_amd: MI355X-native GPTQ/OBQ layer quantization behind the sleekit function surface."""
