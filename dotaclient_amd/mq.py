"""RabbitMQ client with the method surface DotaOptimizer uses (reference: optimizer.py:67-174).

Control plane, outside the hot path: queue `experience` (work queue of pickled rollouts), exchange
`model` (x-recent-history, length 1).  pika is imported lazily so the rest of the package works
without it; tests inject an in-memory object with the same five methods instead.
"""
import logging
import time

logger = logging.getLogger(__name__)


class MessageQueue:
    EXPERIENCE_QUEUE_NAME = 'experience'
    MODEL_EXCHANGE_NAME = 'model'
    MAX_RETRIES = 10

    def __init__(self, host, port, prefetch_count, use_model_exchange):
        import pika
        self._pika = pika
        self._params = pika.ConnectionParameters(host=host, port=port, heartbeat=300)
        self.prefetch_count, self.use_model_exchange = prefetch_count, use_model_exchange
        self._conn = self._xp = self._model = None

    def connect(self):
        if self._conn is not None and not self._conn.is_closed:
            return
        for attempt in range(self.MAX_RETRIES):
            try:
                self._conn = self._pika.BlockingConnection(self._params)
                break
            except self._pika.exceptions.AMQPConnectionError:
                logger.error('RMQ connect failed (%d/%d)', attempt + 1, self.MAX_RETRIES)
                time.sleep(5)
        self._xp = self._conn.channel()
        self._xp.basic_qos(prefetch_count=self.prefetch_count)
        self._xp.queue_declare(queue=self.EXPERIENCE_QUEUE_NAME)
        if self.use_model_exchange:
            self._model = self._conn.channel()
            self._model.exchange_declare(exchange=self.MODEL_EXCHANGE_NAME, exchange_type='x-recent-history',
                                         arguments={'x-recent-history-length': 1})

    def process_data_events(self):
        try:
            self._conn.process_data_events()      # heartbeat between epochs (optimizer.py:471)
        except Exception:
            pass

    def _retry(self, fn):
        try:
            return fn()
        except (self._pika.exceptions.ConnectionClosed, self._pika.exceptions.ChannelClosed):
            logger.error('reconnecting to queue')
            self.connect()
            return fn()

    def consume_xp(self):
        def once():
            method, props, body = next(self._xp.consume(queue=self.EXPERIENCE_QUEUE_NAME))
            self._xp.basic_ack(delivery_tag=method.delivery_tag)
            return method, props, body
        return self._retry(once)

    def publish_model(self, msg, hdr):
        if self._model is None:
            return
        self._retry(lambda: self._model.basic_publish(exchange=self.MODEL_EXCHANGE_NAME, routing_key='', body=msg,
                                                      properties=self._pika.BasicProperties(headers=hdr)))

    def close(self):
        if self._conn is not None and self._conn.is_open:
            self._conn.close()
