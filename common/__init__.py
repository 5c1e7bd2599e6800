PRETRAINED_MODEL_PATH = "./checkpoints/i2vgen-xl"
