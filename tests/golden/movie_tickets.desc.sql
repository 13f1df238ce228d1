date_time	datetime
movie_id	varchar(255)
movie_rank	smallint(5) unsigned
movie	varchar(255)
num_tickets	int(11) unsigned
revenue	decimal(24,12)
