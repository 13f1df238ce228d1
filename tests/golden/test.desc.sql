firstName	varchar(100)
lastName	varchar(100)
age	int(11)
eventCode	int(11)
