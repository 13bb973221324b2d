"""The optimizer the reference compiles with: keras.optimizers.Adam(lr=1e-5, clipnorm=0.001)
(bin/train.py:101).  A plain parameter record; the arithmetic is pp_adam_step_clipnorm."""


class Adam(object):
    def __init__(self, lr=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7, clipnorm=0.0, **kwargs):
        self.lr = float(kwargs.get("learning_rate", lr))
        self.beta_1, self.beta_2, self.epsilon, self.clipnorm = float(beta_1), float(beta_2), float(epsilon), float(clipnorm)


adam = Adam
